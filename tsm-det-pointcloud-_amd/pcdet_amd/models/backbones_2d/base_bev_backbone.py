"""BaseBEVBackbone: dense 2-D conv stack on the BEV map (reference pcdet/models/backbones_2d/base_bev_backbone.py:6-112).
Dense GEMM-shaped work -> stock PyTorch-ROCm (MIOpen / hipBLASLt MFMA kernels).  Same module tree / state_dict keys
(`blocks.i.j`, `deblocks.i.j`).  Fork drift handled: accepts the extra kwargs the fork's template
passes, exposes num_bev_features AND num_voxel_neck_features / num_point_features, and writes both
`spatial_features_2d` and the list `encoded_bev_features` the fork's AnchorHeadSingle reads."""
import numpy as np
import torch
import torch.nn as nn


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
            x = self.blocks[i](x)
            stride = int(spatial_features.shape[2] / x.shape[2])
            data_dict['spatial_features_%dx' % stride] = x
            ups.append(self.deblocks[i](x) if len(self.deblocks) > 0 else x)
        if len(ups) > 1:
            x = torch.cat(ups, dim=1)
        elif len(ups) == 1:
            x = ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = self.deblocks[-1](x)
        data_dict['spatial_features_2d'] = x
        data_dict['encoded_bev_features'] = [x]
        return data_dict
