"""SparseConvTensor — the container the reference builds at
pcdet/models/backbones_3d/spconv_backbone.py:151-156 and consumes through `.features`, `.indices`,
`.spatial_shape`, `.batch_size`, `.indice_dict`, `.replace_feature()` (pcdet/utils/spconv_utils.py:28-34)
and `.dense()` (pcdet/models/backbones_2d/map_to_bev/height_compression.py:21).

Data layout in HBM: features fp32 row-major [N, C]; indices int32 [N, 4] = (batch, z, y, x), one 16-byte
record per voxel so every kernel reads a voxel's coordinate with a single dwordx4 load.
"""
import torch

from . import functional as F_


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None, indice_dict=None,
                 benchmark=False, n_valid=None, static_caps=None, **_unused):
        self._features = features
        if indices.dtype != torch.int32:
            indices = indices.int()
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        # static-capacity (hipGraph) mode: rows are a CAPACITY, the live count is the device tensor n_valid (int64[1]);
        # static_caps maps an indice_key to the output-row capacity of that strided conv
        self.n_valid = n_valid
        self.static_caps = static_caps

    # spconv 2.x makes .features read-only and offers replace_feature(); spconv 1.x assigned to it.  The
    # reference supports both (spconv_utils.py:28-34), so both work here.
    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, value):
        self._features = value

    def replace_feature(self, feature):
        new = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid, self.voxel_num,
                               self.indice_dict, self.benchmark, n_valid=self.n_valid, static_caps=self.static_caps)
        return new

    def shadow_copy(self):
        return self.replace_feature(self._features)

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True, channels_last_memory=False):
        """[N,C] -> [B, C, D, H, W] (zeros elsewhere).  channels_first=False returns [B, D, H, W, C].

        channels_last_memory=True keeps the logical [B,C,D,H,W] shape but stores it as [B,H,W,C,D], so the
        reference's `.view(N, C*D, H, W)` (height_compression.py:22-23) yields a channels_last BEV map.
        """
        out = F_.dense(self._features, self.indices, self.batch_size, self.spatial_shape, channels_last_memory,
                       self.n_valid)
        if not channels_first:
            return out.permute(0, 2, 3, 4, 1).contiguous()
        return out

    @property
    def sparity(self):  # (sic) spconv's spelling
        return self.indices.shape[0] / float(self.spatial_size * self.batch_size)

    @property
    def device(self):
        return self._features.device

    def __repr__(self):
        return "SparseConvTensor[shape=%s, spatial=%s, batch=%d]" % (tuple(self._features.shape), self.spatial_shape,
                                                                      self.batch_size)
