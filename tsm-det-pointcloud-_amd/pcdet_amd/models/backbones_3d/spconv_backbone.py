"""VoxelBackBone8x on libspx — same constructor, state_dict keys, batch_dict keys and attributes as the
reference's pcdet/models/backbones_3d/spconv_backbone.py:77-194 (stage wiring :85-125, outputs :169-192).

The 12 sparse convolutions (8 SubM k3, 3 strided k3 s2, 1 k(3,1,1) s(2,1,1)) run as hand-written HIP kernels;
BatchNorm1d(eps=1e-3, momentum=0.01) + ReLU stay torch.nn modules so parameter names (`conv2.0.1.weight` ...)
and train/eval behaviour are unchanged.
"""
import os
from functools import partial

import torch
import torch.nn as nn

import spx as spconv
from spx.functional import bn_act, pack_all
from spx.prebuild import prebuild

from ...utils.spconv_utils import replace_feature  # noqa: F401  (API parity)


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm',
                   norm_fn=None, active=True):
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError
    layers = [conv, norm_fn(out_channels)]
    if active:
        layers.append(nn.ReLU())
    return spconv.SparseSequential(*layers)


class SparseBasicBlock(spconv.SparseModule):
    """Residual block of two submanifold convs sharing one rulebook (reference spconv_backbone.py:38-74; SURVEY §8 row
    f-3).  bias=True on both convs, BN after each, identity (or `downsample`) added before the last ReLU."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=True,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=True,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.conv1(x)
        out = out.replace_feature(bn_act(out.features, self.bn1, True, d_n=out.n_valid))
        out = self.conv2(out)
        # bn2 + identity + ReLU: one fused kernel pair in training (spx_bn_add_relu_*), the torch modules otherwise
        out = out.replace_feature(bn_act(out.features, self.bn2, True, identity.features, d_n=out.n_valid))
        return out


_AHEAD = os.environ.get("SPX_TABLES_AHEAD", "1") != "0"      # dev knob


class _TablesAhead(object):
    """Launch the strided rule table of the NEXT stage early, collect it late (see _run_8x_stack)."""

    def __init__(self, x):
        self.on = (_AHEAD and x.features.is_cuda and x.n_valid is None and x.indices.shape[0] > 0
                   and not torch.cuda.is_current_stream_capturing())
        self.book, self.batch = x.indice_dict, x.batch_size
        self.pending = None

    @staticmethod
    def _first_strided(stage):
        for m in stage.modules():
            if isinstance(m, spconv.SparseConvolution):
                return m if (not m.subm and not m.inverse and m.indice_key is not None) else None
        return None

    def launch(self, stage, indices, shape):
        m = self._first_strided(stage) if (self.on and stage is not None) else None
        if m is None or m.indice_key in self.book or indices.shape[0] == 0:
            self.pending = None
            return
        self.pending = (m.indice_key, spconv.ops.conv_rulebook(indices, self.batch, shape, m.kernel_size, m.stride,
                                                               m.padding, m.dilation, sync="later"))

    def collect_and_launch(self, next_stage):
        if self.pending is None:
            return
        key, pend = self.pending
        rb = self.book[key] = pend.finish()
        self.launch(next_stage, rb.out_indices, rb.out_shape)


def _queue_tables(self, x):
    """Static-capacity tensors: every rule table of the stack (+ grouping and work plans) goes to the index stream now and
    is built in the shadow of the convolutions (spx/prebuild.py); exact-size tensors build theirs lazily, as before."""
    convs = [m for m in self.modules() if isinstance(m, spconv.SparseConvolution)]
    if x.features.is_cuda and torch.is_grad_enabled():
        pack_all(convs)              # all packed weight copies of the step in one launch (they all changed in optimizer.step)
    if x.n_valid is not None:
        prebuild(x, convs)


def _run_8x_stack(self, batch_dict, with_points_keys):
    voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
    x = spconv.SparseConvTensor(features=voxel_features, indices=voxel_coords.int(), spatial_shape=self.sparse_shape,
                                batch_size=batch_dict['batch_size'], n_valid=batch_dict.get('voxel_num_valid', None),
                                static_caps=batch_dict.get('static_caps', None))
    _queue_tables(self, x)
    # A strided rule table has to tell the host its row count.  Each one is LAUNCHED one stage early — its kernels sit in
    # the stream in front of the previous stage's convolutions — and its count is picked up after that stage has been
    # queued, so the host never waits on an empty GPU (same tables, found by the modules under their indice_key).
    ahead = _TablesAhead(x)
    ahead.launch(self.conv2, x.indices, x.spatial_shape)
    x = self.conv_input(x)
    x_conv1 = self.conv1(x)
    ahead.collect_and_launch(self.conv3)
    x_conv2 = self.conv2(x_conv1)
    ahead.collect_and_launch(self.conv4)
    x_conv3 = self.conv3(x_conv2)
    ahead.collect_and_launch(self.conv_out)
    x_conv4 = self.conv4(x_conv3)
    ahead.collect_and_launch(None)
    out = self.conv_out(x_conv4)
    batch_dict['encoded_spconv_tensor'] = out
    batch_dict['encoded_spconv_tensor_stride'] = 8
    feats = {'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4}
    strides = {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}
    if with_points_keys:
        feats.update({'x_points_mean': x_conv2, 'x_points_max': x_conv2})
        strides.update({'x_points_mean': 2, 'x_points_max': 2})
    batch_dict['multi_scale_3d_features'] = feats
    batch_dict['multi_scale_3d_strides'] = strides
    return batch_dict


class VoxelResBackBone8x(nn.Module):
    """Residual variant (reference spconv_backbone.py:197-307): same stage layout, SparseBasicBlock pairs, 128-wide
    stage 4.  Same kernels as VoxelBackBone8x; parameter names follow the reference (`conv1.0.conv1.weight`, ...)."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(grid_size[2]) + 1, int(grid_size[1]), int(grid_size[0])]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        block = partial(post_act_block, norm_fn=norm_fn)
        res = partial(SparseBasicBlock, norm_fn=norm_fn)
        self.conv1 = spconv.SparseSequential(res(16, 16, indice_key='res1'), res(16, 16, indice_key='res1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            res(32, 32, indice_key='res2'), res(32, 32, indice_key='res2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            res(64, 64, indice_key='res3'), res(64, 64, indice_key='res3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 128, 3, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            res(128, 128, indice_key='res4'), res(128, 128, indice_key='res4'))
        last_pad = self.model_cfg.get('last_pad', 0) if hasattr(self.model_cfg, 'get') else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 128}

    def forward(self, batch_dict):
        return _run_8x_stack(self, batch_dict, with_points_keys=False)


class VoxelBackBone8x(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        # (z, y, x) order with one extra z cell: 40 -> 41 so that three stride-2 stages give 41->21->11->5
        self.sparse_shape = [int(grid_size[2]) + 1, int(grid_size[1]), int(grid_size[0])]

        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'),
            norm_fn(16),
            nn.ReLU(),
        )
        block = partial(post_act_block, norm_fn=norm_fn)
        self.conv1 = spconv.SparseSequential(
            block(16, 16, 3, padding=1, indice_key='subm1'),
        )
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, padding=1, indice_key='subm2'),
            block(32, 32, 3, padding=1, indice_key='subm2'),
        )
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, padding=1, indice_key='subm3'),
            block(64, 64, 3, padding=1, indice_key='subm3'),
        )
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            block(64, 64, 3, padding=1, indice_key='subm4'),
            block(64, 64, 3, padding=1, indice_key='subm4'),
        )
        last_pad = self.model_cfg.get('last_pad', 0) if hasattr(self.model_cfg, 'get') else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'),
            norm_fn(128),
            nn.ReLU(),
        )
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64, 'x_points_mean': 32,
                                  'x_points_max': 32}

    def forward(self, batch_dict):
        """batch_dict in: voxel_features [N,C], voxel_coords [N,4] (b,z,y,x), batch_size.
        out: encoded_spconv_tensor (+_stride 8), multi_scale_3d_features / _strides."""
        voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        x = spconv.SparseConvTensor(features=voxel_features, indices=voxel_coords.int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'],
                                    n_valid=batch_dict.get('voxel_num_valid', None),
                                    static_caps=batch_dict.get('static_caps', None))
        _queue_tables(self, x)
        x = self.conv_input(x)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        out = self.conv_out(x_conv4)  # [200, 176, 5] -> [200, 176, 2]

        batch_dict['encoded_spconv_tensor'] = out
        batch_dict['encoded_spconv_tensor_stride'] = 8
        batch_dict['multi_scale_3d_features'] = {'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3,
                                                 'x_conv4': x_conv4, 'x_points_mean': x_conv2,
                                                 'x_points_max': x_conv2}
        batch_dict['multi_scale_3d_strides'] = {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8,
                                                'x_points_mean': 2, 'x_points_max': 2}
        return batch_dict
