"""HeightCompression: BEV collapse (reference pcdet/models/backbones_2d/map_to_bev/height_compression.py:4-26).
.dense() is libspx's densify kernel; with CHANNELS_LAST (default True) the dense tensor is laid out so that
the .view(N, C*D, H, W) below is already channels_last for the 2-D backbone — same values, no copy."""
import torch.nn as nn


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.channels_last = bool(model_cfg.get('CHANNELS_LAST', True)) if hasattr(model_cfg, 'get') else True

    def forward(self, batch_dict):
        encoded = batch_dict['encoded_spconv_tensor']
        dense = encoded.dense(channels_last_memory=self.channels_last)
        n, c, d, h, w = dense.shape
        bev = dense.view(n, c * d, h, w)                               # BEV channel = c * D + z
        # where this map came from: lets BaseBEVBackbone run its first convolution over the active rows only (the map
        # is zero everywhere else); a tensor derived from `bev` does not carry the tag
        bev._spx_source = (encoded, bev._version)
        batch_dict['spatial_features'] = bev
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
