"""BaseBEVBackbone: dense 2-D conv stack on the BEV map (reference pcdet/models/backbones_2d/base_bev_backbone.py:6-112).
Dense GEMM-shaped work -> stock PyTorch-ROCm (MIOpen / hipBLASLt MFMA kernels).  Same module tree / state_dict keys
(`blocks.i.j`, `deblocks.i.j`).  Fork drift handled: accepts the extra kwargs the fork's template
passes, exposes num_bev_features AND num_voxel_neck_features / num_point_features, and writes both
`spatial_features_2d` and the list `encoded_bev_features` the fork's AnchorHeadSingle reads."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from spx import ops
from spx.functional import (bn_relu_cat_train, bn_relu_train, dense as densify_rows, folded_bn, sparse_conv, wino_conv2d,
                            wino_conv2d_ok, wino_conv_bn_relu_train)

# below this many activations the three-launch fused BN kernel is launch-bound and MIOpen's BN is faster (measured on
# MI355X, tools/dense_tail_probe.py: 18 M elements 176 -> 132 us fwd+bwd, 9 M elements 98 -> 123 us)
# measured (round 2, alternating runs on one box): 8 M instead of 12 M elements brings the 256-channel KITTI maps (9.0 M)
# in: +1.4 % of the cfg-2 step; 4 M / 2 M make no difference on the Waymo maps
_FUSED_BN_MIN_ELEMS = int(os.environ.get("SPX_BEV_FUSED_BN_MIN", str(8 * 1000 * 1000)))


def _run_block(seq, x, start=0):
    """nn.Sequential.forward (from module `start` on) with two rewrites that leave every value unchanged:
    * ZeroPad2d(1) + Conv2d(padding=0)  ->  the same conv with padding=1 (no padded copy of the BEV map);
    * training BatchNorm2d + ReLU on a channels_last map -> libspx's fused BN+ReLU over the [B*H*W, C] row view
      (the same kernels the sparse backbone uses; one pass less forward, two less backward);
    * Conv2d(c, c', 3, padding=1, bias=False) on a channels_last fp32 map -> libspx's Winograd F(2x2, 3x3) kernel (forward
      and data gradient; 2.25x fewer fp32 multiplies than the vendor's direct form, same fp32 arithmetic), with the
      inference BatchNorm2d + ReLU that follow folded into its epilogue."""
    mods = list(seq)
    i = start
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if (isinstance(m, nn.ZeroPad2d) and isinstance(nxt, nn.Conv2d) and tuple(nxt.padding) == (0, 0)
                and nxt.padding_mode == 'zeros' and len(set(m.padding)) == 1 and isinstance(nxt.padding, tuple)):
            p = int(m.padding[0])
            x = F.conv2d(x, nxt.weight, nxt.bias, nxt.stride, (p, p), nxt.dilation, nxt.groups)
            i += 2
            continue
        if _WINO and isinstance(m, nn.Conv2d) and wino_conv2d_ok(x, m):
            nn2 = mods[i + 2] if i + 2 < len(mods) else None
            if (_CONV_BN_NODE and isinstance(nxt, nn.BatchNorm2d) and isinstance(nn2, nn.ReLU) and _train_bn_ok(nxt)
                    and torch.is_grad_enabled()
                    and x.shape[0] * x.shape[2] * x.shape[3] * m.out_channels >= _FUSED_BN_MIN_ELEMS
                    and 1024 % m.out_channels == 0):
                # training: conv + BN + ReLU as one node; the conv epilogue takes the BN statistics while it stores y
                x = wino_conv_bn_relu_train(x, m, nxt)
                i += 3
            elif nxt is not None and isinstance(nn2, nn.ReLU) and _eval_bn_ok(nxt, x, check_layout=False):
                scale, shift, _ = folded_bn(nxt)
                x = wino_conv2d(x, m.weight, scale=scale, shift=shift, relu=True)
                i += 3
            else:
                x = wino_conv2d(x, m.weight)
                i += 1
            continue
        if (isinstance(m, nn.BatchNorm2d) and isinstance(nxt, nn.ReLU) and m.training and m.affine
                and m.track_running_stats and m.momentum is not None and x.is_cuda and x.dtype == torch.float32
                and x.numel() >= _FUSED_BN_MIN_ELEMS and 1024 % x.shape[1] == 0 and x.shape[1] % 4 == 0
                and x.is_contiguous(memory_format=torch.channels_last)):
            b, c, h, w = x.shape
            y = bn_relu_train(x.permute(0, 2, 3, 1).reshape(b * h * w, c), m, True)
            x = y.view(b, h, w, c).permute(0, 3, 1, 2)
            i += 2
            continue
        if _eval_bn_ok(m, x) and isinstance(nxt, nn.ReLU):
            b, c, h, w = x.shape
            y = _bn_eval_rows(x.permute(0, 2, 3, 1).reshape(b * h * w, c), m, True)
            x = y.view(b, h, w, c).permute(0, 3, 1, 2)
            i += 2
            continue
        x = m(x)
        i += 1
    return x


_FUSED_EVAL_BN = os.environ.get("SPX_BEV_FUSED_EVAL_BN", "1") != "0"       # dev knob


_WINO = os.environ.get("SPX_BEV_WINOGRAD", "1") != "0"                      # dev knob
_CONV_BN_NODE = os.environ.get("SPX_BEV_CONV_BN_NODE", "1") != "0"         # dev knob


def _train_bn_ok(m):
    """BatchNorm2d in training mode that libspx's fused kernels cover (affine, running statistics, fixed momentum)."""
    return m.training and m.affine and m.track_running_stats and m.momentum is not None and m.weight.dtype == torch.float32


def _eval_bn_ok(m, x, check_layout=True):
    """Inference: BatchNorm2d on running statistics + ReLU over a channels_last fp32 map = ONE libspx pass over its
    [B*H*W, C] rows (spx_bn_apply) instead of two torch elementwise passes."""
    return (_FUSED_EVAL_BN and isinstance(m, nn.BatchNorm2d) and not m.training and m.affine and m.running_mean is not None
            and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and not torch.is_autocast_enabled()
            and (not check_layout or (1024 % x.shape[1] == 0 and x.shape[1] % 4 == 0
                                      and x.is_contiguous(memory_format=torch.channels_last))))


def _bn_eval_rows(rows, bn, relu, out=None):
    _scale, _shift, invstd = folded_bn(bn)
    return ops.bn_apply(rows, bn.running_mean, invstd, bn.weight, bn.bias, relu, out=out)


_SPARSE_ENTRY = os.environ.get("SPX_BEV_SPARSE_ENTRY", "1") != "0"     # dev knob
_FUSED_CAT = os.environ.get("SPX_BEV_FUSED_CAT", "1") != "0"            # dev knob
_WIDE_BALANCED = os.environ.get("SPX_BEV_WIDE_BALANCED", "1") != "0"    # dev knob
_SPARSE_ENTRY_CHANNELS = (16, 32, 64, 128)                                # MFMA instantiations of libspx's conv kernels


def _sparse_entry(seq, bev):
    """First convolution of a block over the ACTIVE rows of the sparse tensor the BEV map was densified from.

    `bev` = HeightCompression's view [B, C*D, H, W] of encoded.dense() (reference height_compression.py:21-23) is zero
    outside the encoded tensor's rows (12 % of the pixels in the KITTI configuration), and ZeroPad2d(p) + Conv2d(C*D, Co,
    k, stride s, bias=False) over it (reference base_bev_backbone.py:27-34) is term by term the sparse convolution of
    the encoded tensor with kernel (D, k, k), stride (1, s, s), padding (0, p, p) — z folds into the channel index
    c*D + z exactly as the view does — followed by densification: every product the dense conv adds beyond those is
    an exact zero.  ~8x fewer FLOPs forward, and the backward pass produces the gradient of the active rows only.
    Returns the conv output [B, Co, Ho, Wo] (same memory format as `bev`), or None when the pattern does not apply."""
    tag = getattr(bev, '_spx_source', None)
    mods = list(seq)
    if not _SPARSE_ENTRY or tag is None or tag[1] != bev._version or len(mods) < 2:     # untagged, or written to since
        return None
    src = tag[0]
    pad, conv = mods[0], mods[1]
    if not (isinstance(pad, nn.ZeroPad2d) and isinstance(conv, nn.Conv2d) and len(set(pad.padding)) == 1
            and tuple(conv.padding) == (0, 0) and conv.padding_mode == 'zeros' and conv.bias is None and conv.groups == 1
            and tuple(conv.dilation) == (1, 1) and conv.stride[0] == conv.stride[1]):
        return None
    feats = src.features
    c, d = feats.shape[1], int(src.spatial_shape[0])
    kh, kw = conv.kernel_size
    if not (feats.is_cuda and feats.dtype == torch.float32 and bev.dtype == torch.float32
            and not torch.is_autocast_enabled() and conv.in_channels == c * d and d * kh * kw <= 30
            and c in _SPARSE_ENTRY_CHANNELS and conv.out_channels in _SPARSE_ENTRY_CHANNELS and feats.shape[0] > 0):
        return None
    p, s = int(pad.padding[0]), int(conv.stride[0])
    if src.n_valid is None:
        if torch.cuda.is_current_stream_capturing():
            return None                                  # exact-size tables need a host sync
        rb = ops.conv_rulebook(src.indices, src.batch_size, src.spatial_shape, (d, kh, kw), (1, s, s), (0, p, p))
    else:
        # static-capacity tensors (hipGraph inference): rows at capacity, live counts stay on the device, no sync
        rb = ops.conv_rulebook(src.indices, src.batch_size, src.spatial_shape, (d, kh, kw), (1, s, s), (0, p, p),
                               d_n_in=src.n_valid, cap=(src.static_caps or {}).get('bev_entry', None), sync=False)
    if rb.n_out == 0:
        return None
    rb.wide_balanced_fwd = _WIDE_BALANCED   # 128 -> 128 forward through the two-half balanced kernel (spx/ops.py: balanced_ok)
    # Conv2d weight [Co, c*D + z, ky, kx] -> sparse layout [Co, z, ky, kx, c]; a view, so the gradient lands in conv.weight
    w3 = conv.weight.view(conv.out_channels, c, d, kh, kw).permute(0, 2, 3, 4, 1)
    rows = sparse_conv(feats, w3, None, rb)
    channels_last = bev.is_contiguous(memory_format=torch.channels_last)
    out = densify_rows(rows, rb.out_indices, src.batch_size, rb.out_shape, channels_last, rb.d_n_out)   # [B, Co, 1, Ho, Wo]
    return out.view(out.shape[0], out.shape[1], out.shape[3], out.shape[4])


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

    def _cat_fusable(self):
        """The up-sampling branches end in training-mode BatchNorm2d + ReLU and are concatenated right after: their
        BN+ReLU kernels can write into the channel slices of the concatenated map (no torch.cat copy forward, no slice
        copies backward)."""
        if not _FUSED_CAT or len(self.deblocks) != len(self.blocks) or len(self.deblocks) < 2:
            return False
        for d in self.deblocks:
            mods = list(d)
            if not (len(mods) == 3 and isinstance(mods[1], nn.BatchNorm2d) and isinstance(mods[2], nn.ReLU)
                    and mods[1].training and mods[1].affine and mods[1].track_running_stats and mods[1].momentum is not None
                    and 1024 % mods[1].num_features == 0 and mods[1].num_features % 4 == 0):
                return False
        return True

    def _cat_eval_fusable(self, x):
        """Inference twin of _cat_fusable: every up-sampling branch is conv, BatchNorm2d (running statistics), ReLU."""
        if not _FUSED_EVAL_BN or len(self.deblocks) != len(self.blocks) or len(self.deblocks) < 2 or torch.is_grad_enabled():
            return False
        return all(len(d) == 3 and isinstance(d[1], nn.BatchNorm2d) and isinstance(d[2], nn.ReLU) and not d[1].training
                   and d[1].affine and d[1].running_mean is not None and 1024 % d[1].num_features == 0
                   and d[1].num_features % 4 == 0 for d in self.deblocks)

    def forward(self, data_dict):
        spatial_features = data_dict['spatial_features']
        ups, pre = [], []
        x = spatial_features
        plain = x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
        fuse_cat = self._cat_fusable() and plain
        fuse_cat_eval = (not fuse_cat) and plain and self._cat_eval_fusable(x)
        for i in range(len(self.blocks)):
            y = _sparse_entry(self.blocks[i], x) if i == 0 else None
            x = _run_block(self.blocks[i], x) if y is None else _run_block(self.blocks[i], y, start=2)
            stride = int(spatial_features.shape[2] / x.shape[2])
            data_dict['spatial_features_%dx' % stride] = x
            if fuse_cat or fuse_cat_eval:
                pre.append(self.deblocks[i][0](x))                      # the up-sampling conv; BN + ReLU follow below
            else:
                ups.append(_run_block(self.deblocks[i], x) if len(self.deblocks) > 0 else x)
        if fuse_cat or fuse_cat_eval:
            b, _c, h, w = pre[0].shape
            same = all(t.shape[0] == b and t.shape[2:] == pre[0].shape[2:]
                       and t.is_contiguous(memory_format=torch.channels_last) for t in pre)
            if same and fuse_cat and all(t.numel() >= _FUSED_BN_MIN_ELEMS for t in pre):
                rows = [t.permute(0, 2, 3, 1).reshape(b * h * w, t.shape[1]) for t in pre]
                y = bn_relu_cat_train(rows, [d[1] for d in self.deblocks])
                ups = [y.view(b, h, w, y.shape[1]).permute(0, 3, 1, 2)]
            elif same and fuse_cat_eval:
                # inference: BN (running statistics) + ReLU of every branch in one pass each, written straight into its
                # channel slice of the concatenated map (no torch.cat copy)
                widths = [t.shape[1] for t in pre]
                y = torch.empty((b * h * w, sum(widths)), dtype=torch.float32, device=x.device)
                off = 0
                for t, d, wd in zip(pre, self.deblocks, widths):
                    _bn_eval_rows(t.permute(0, 2, 3, 1).reshape(b * h * w, wd), d[1], True, out=y[:, off:off + wd])
                    off += wd
                ups = [y.view(b, h, w, y.shape[1]).permute(0, 3, 1, 2)]
            else:
                ups = [_run_block(d, t, start=1) for d, t in zip(self.deblocks, pre)]
        if len(ups) > 1:
            x = torch.cat(ups, dim=1)
        elif len(ups) == 1:
            x = ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = _run_block(self.deblocks[-1], x)
        data_dict['spatial_features_2d'] = x
        data_dict['encoded_bev_features'] = [x]
        return data_dict
