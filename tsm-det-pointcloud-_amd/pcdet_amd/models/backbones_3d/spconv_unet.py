"""UNetV2 — the sparse encoder-decoder of Part-A2 on libspx (reference
pcdet/models/backbones_3d/spconv_unet.py:49-212).

Same constructor, parameter names (`conv_up_t4.conv1.weight`, `inv_conv3.0.weight`, ...) and batch_dict contract as the
reference: the encoder is the VoxelBackBone8x stage layout, each decoder level k runs
    lateral -> SparseBasicBlock (`conv_up_t{k}`) ; concat with the level below ; SubM conv (`conv_up_m{k}`) ;
    + channel-folded skip (reference :146-160) ; SparseInverseConv3d back to level k-1 (`inv_conv{k}`, rulebook of
    `spconv{k}` with the two tables swapped; the last level uses a SubM conv `conv5` instead)
and the full-resolution features come back as `point_features` with their voxel centres in `raw_points_bxyz`.
Nothing here is a new kernel: SubM / strided / inverse convolutions are the libspx kernels of the main path (the
128->64 `conv_up_m` layers use the 128-channel MFMA instantiations).
"""
from functools import partial

import torch
import torch.nn as nn

import spx as spconv
from spx.functional import bn_act

from ...utils import common_utils
from .spconv_backbone import post_act_block


class SparseBasicBlock(spconv.SparseModule):
    """Residual pair of bias-free SubM convs sharing one rulebook (reference spconv_unet.py:11-46)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, indice_key=None, norm_fn=None):
        super().__init__()
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        assert x.features.dim() == 2, 'x.features.dim()=%d' % x.features.dim()
        skip = x.features if self.downsample is None else self.downsample(x)
        y = self.conv1(x)
        y = y.replace_feature(bn_act(y.features, self.bn1, True))
        y = self.conv2(y)
        return y.replace_feature(bn_act(y.features, self.bn2, True, skip))          # bn2 + skip + ReLU, one kernel pair


# encoder stages: (name, [(cin, cout, conv_type, indice_key, stride, padding), ...])
_ENCODER = (
    ('conv1', [(16, 16, 'subm', 'subm1', 1, 1)]),
    ('conv2', [(16, 32, 'spconv', 'spconv2', 2, 1), (32, 32, 'subm', 'subm2', 1, 1), (32, 32, 'subm', 'subm2', 1, 1)]),
    ('conv3', [(32, 64, 'spconv', 'spconv3', 2, 1), (64, 64, 'subm', 'subm3', 1, 1), (64, 64, 'subm', 'subm3', 1, 1)]),
    ('conv4', [(64, 64, 'spconv', 'spconv4', 2, (0, 1, 1)), (64, 64, 'subm', 'subm4', 1, 1),
               (64, 64, 'subm', 'subm4', 1, 1)]),
)
# decoder levels: (level, lateral channels, merged channels out, channels after the inverse conv, subm key, spconv key)
_DECODER = ((4, 64, 64, 64, 'subm4', 'spconv4'), (3, 64, 64, 32, 'subm3', 'spconv3'), (2, 32, 32, 16, 'subm2', 'spconv2'))


class UNetV2(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, voxel_size, point_cloud_range, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.sparse_shape = [int(grid_size[2]) + 1, int(grid_size[1]), int(grid_size[0])]
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        block = partial(post_act_block, norm_fn=norm_fn)

        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        for name, layers in _ENCODER:
            setattr(self, name, spconv.SparseSequential(*[
                block(ci, co, 3, stride=st, padding=pd, indice_key=key, conv_type=ct) for ci, co, ct, key, st, pd in layers]))

        cfg_get = self.model_cfg.get if hasattr(self.model_cfg, 'get') else (lambda k, d=None: d)
        if cfg_get('RETURN_ENCODED_TENSOR', True):
            self.conv_out = spconv.SparseSequential(
                spconv.SparseConv3d(64, 64, (1, 1, 1), stride=(1, 1, 1), padding=cfg_get('last_pad', 0), bias=False,
                                    indice_key='spconv_down2'), norm_fn(64), nn.ReLU())
        else:
            self.conv_out = None

        for lvl, c_lat, c_mid, c_up, subm_key, sp_key in _DECODER:
            setattr(self, 'conv_up_t%d' % lvl, SparseBasicBlock(c_lat, c_lat, indice_key=subm_key, norm_fn=norm_fn))
            setattr(self, 'conv_up_m%d' % lvl, block(2 * c_lat, c_mid, 3, padding=1, indice_key=subm_key))
            setattr(self, 'inv_conv%d' % lvl, block(c_mid, c_up, 3, indice_key=sp_key, conv_type='inverseconv'))
        self.conv_up_t1 = SparseBasicBlock(16, 16, indice_key='subm1', norm_fn=norm_fn)
        self.conv_up_m1 = block(32, 16, 3, indice_key='subm1')
        self.conv5 = spconv.SparseSequential(block(16, 16, 3, padding=1, indice_key='subm1'))
        self.num_point_features = 16

    @staticmethod
    def channel_reduction(x, out_channels):
        """[N, C1] -> [N, C2] by summing groups of C1/C2 consecutive channels (reference :146-160)."""
        feats = x.features
        n, c1 = feats.shape
        assert c1 % out_channels == 0 and c1 >= out_channels
        return x.replace_feature(feats.view(n, out_channels, -1).sum(dim=2))

    def UR_block_forward(self, x_lateral, x_bottom, conv_t, conv_m, conv_inv):
        x_t = conv_t(x_lateral)
        merged = x_t.replace_feature(torch.cat((x_bottom.features, x_t.features), dim=1))
        x_m = conv_m(merged)
        folded = self.channel_reduction(merged, x_m.features.shape[1])
        return conv_inv(folded.replace_feature(x_m.features + folded.features))

    def forward(self, batch_dict):
        x = spconv.SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        enc = {}
        for name, _ in _ENCODER:
            x = getattr(self, name)(x)
            enc[name] = x
        if self.conv_out is not None:   # detection branch
            batch_dict['encoded_spconv_tensor'] = self.conv_out(enc['conv4'])
            batch_dict['encoded_spconv_tensor_stride'] = 8

        up = enc['conv4']               # segmentation branch: decode 4 -> 1
        for lvl in (4, 3, 2):
            up = self.UR_block_forward(enc['conv%d' % lvl], up, getattr(self, 'conv_up_t%d' % lvl),
                                       getattr(self, 'conv_up_m%d' % lvl), getattr(self, 'inv_conv%d' % lvl))
        up = self.UR_block_forward(enc['conv1'], up, self.conv_up_t1, self.conv_up_m1, self.conv5)

        batch_dict['point_features'] = up.features
        centers = common_utils.get_voxel_centers(up.indices[:, 1:], downsample_times=1, voxel_size=self.voxel_size,
                                                 point_cloud_range=self.point_cloud_range)
        batch_dict['raw_points_bxyz'] = torch.cat((up.indices[:, 0:1].float(), centers), dim=1)
        return batch_dict
