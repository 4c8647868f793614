"""BaseBEVBackbone: dense 2-D conv stack on the BEV map (reference pcdet/models/backbones_2d/base_bev_backbone.py:6-112).
Dense GEMM-shaped work -> stock PyTorch-ROCm (MIOpen / hipBLASLt MFMA kernels).  Same module tree / state_dict keys
(`blocks.i.j`, `deblocks.i.j`).  Fork drift handled: accepts the extra kwargs the fork's template
passes, exposes num_bev_features AND num_voxel_neck_features / num_point_features, and writes both
`spatial_features_2d` and the list `encoded_bev_features` the fork's AnchorHeadSingle reads."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from spx.functional import bn_relu_train

# below this many activations the three-launch fused BN kernel is launch-bound and MIOpen's BN is faster (measured on
# MI355X, tools/dense_tail_probe.py: 18 M elements 176 -> 132 us fwd+bwd, 9 M elements 98 -> 123 us)
_FUSED_BN_MIN_ELEMS = 12 * 1024 * 1024


def _run_block(seq, x):
    """nn.Sequential.forward with two rewrites that leave every value unchanged:
    * ZeroPad2d(1) + Conv2d(padding=0)  ->  the same conv with padding=1 (no padded copy of the BEV map);
    * training BatchNorm2d + ReLU on a channels_last map -> libspx's fused BN+ReLU over the [B*H*W, C] row view
      (the same kernels the sparse backbone uses; one pass less forward, two less backward)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if (isinstance(m, nn.ZeroPad2d) and isinstance(nxt, nn.Conv2d) and tuple(nxt.padding) == (0, 0)
                and nxt.padding_mode == 'zeros' and len(set(m.padding)) == 1 and isinstance(nxt.padding, tuple)):
            p = int(m.padding[0])
            x = F.conv2d(x, nxt.weight, nxt.bias, nxt.stride, (p, p), nxt.dilation, nxt.groups)
            i += 2
            continue
        if (isinstance(m, nn.BatchNorm2d) and isinstance(nxt, nn.ReLU) and m.training and m.affine
                and m.track_running_stats and x.is_cuda and x.dtype == torch.float32
                and x.numel() >= _FUSED_BN_MIN_ELEMS and 1024 % x.shape[1] == 0 and x.shape[1] % 4 == 0
                and x.is_contiguous(memory_format=torch.channels_last)):
            b, c, h, w = x.shape
            y = bn_relu_train(x.permute(0, 2, 3, 1).reshape(b * h * w, c), m, True)
            x = y.view(b, h, w, c).permute(0, 3, 1, 2)
            i += 2
            continue
        x = m(x)
        i += 1
    return x


class BaseBEVBackbone(nn.Module):
    def __init__(self, model_cfg, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = list(model_cfg.get('LAYER_NUMS', None) or [])
        layer_strides = list(model_cfg.get('LAYER_STRIDES', None) or [])
        num_filters = list(model_cfg.get('NUM_FILTERS', None) or [])
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        upsample_strides = list(model_cfg.get('UPSAMPLE_STRIDES', None) or [])
        num_upsample_filters = list(model_cfg.get('NUM_UPSAMPLE_FILTERS', None) or [])
        assert len(upsample_strides) == len(num_upsample_filters)

        def bn(c):
            return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)

        num_levels = len(layer_nums)
        c_in_list = [input_channels, *num_filters[:-1]]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx in range(num_levels):
            c_out = num_filters[idx]
            layers = [nn.ZeroPad2d(1),
                      nn.Conv2d(c_in_list[idx], c_out, kernel_size=3, stride=layer_strides[idx], padding=0, bias=False),
                      bn(c_out), nn.ReLU()]
            for _ in range(layer_nums[idx]):
                layers += [nn.Conv2d(c_out, c_out, kernel_size=3, padding=1, bias=False), bn(c_out), nn.ReLU()]
            self.blocks.append(nn.Sequential(*layers))
            if len(upsample_strides) > 0:
                stride = upsample_strides[idx]
                if stride >= 1:
                    up = nn.ConvTranspose2d(c_out, num_upsample_filters[idx], stride, stride=stride, bias=False)
                else:
                    s = int(np.round(1 / stride))
                    up = nn.Conv2d(c_out, num_upsample_filters[idx], s, stride=s, bias=False)
                self.deblocks.append(nn.Sequential(up, bn(num_upsample_filters[idx]), nn.ReLU()))

        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > num_levels:
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_in, c_in, upsample_strides[-1], stride=upsample_strides[-1], bias=False),
                bn(c_in), nn.ReLU()))
        self.num_bev_features = c_in
        self.num_voxel_neck_features = c_in            # read by the fork's build_backbone_2d
        self.num_point_features = kwargs.get('num_point_features', None)

    def forward(self, data_dict):
        spatial_features = data_dict['spatial_features']
        ups = []
        x = spatial_features
        for i in range(len(self.blocks)):
            x = _run_block(self.blocks[i], x)
            stride = int(spatial_features.shape[2] / x.shape[2])
            data_dict['spatial_features_%dx' % stride] = x
            ups.append(_run_block(self.deblocks[i], x) if len(self.deblocks) > 0 else x)
        if len(ups) > 1:
            x = torch.cat(ups, dim=1)
        elif len(ups) == 1:
            x = ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = _run_block(self.deblocks[-1], x)
        data_dict['spatial_features_2d'] = x
        data_dict['encoded_bev_features'] = [x]
        return data_dict
