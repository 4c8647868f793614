"""spconv.pytorch-compatible module surface (SURVEY.md §8b boundary B2) on top of libspx.

Names, constructor arguments, parameter names/shapes and indice_key caching follow what the reference uses:
  spconv.SparseSequential / SparseModule  — pcdet/models/backbones_3d/spconv_backbone.py:24-33,38,85-125
  spconv.SubMConv3d / SparseConv3d / SparseInverseConv3d — spconv_backbone.py:13-19,46-53,86,121-122
  spconv.conv.SparseConvolution (isinstance target of find_all_spconv_keys) — pcdet/utils/spconv_utils.py:19
  weight [Cout, kz, ky, kx, Cin], optional bias [Cout] — detector3d_template.py:547-562
"""
import math
from collections import OrderedDict

import os

import weakref

import torch
from torch import nn

from . import functional as F_
from . import ops
from . import prebuild as _prebuild
from .tensor import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: modules that take and return a SparseConvTensor."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    return isinstance(module, SparseConvolution)


class SparseSequential(SparseModule):
    """Sequential container that routes SparseConvTensors to sparse modules and `.features` to dense ones
    (BatchNorm1d, ReLU, ...), as the reference relies on at spconv_backbone.py:24-33."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                fuse = _fusable_tail(module, mods, i) if isinstance(input, SparseConvTensor) else None
                if fuse is not None:
                    # inference: BatchNorm1d (running stats) + ReLU folded into the conv kernel's epilogue —
                    # one launch instead of three and no extra pass over the features
                    scale, shift, relu, consumed = fuse
                    input = module(input, fused=(scale, shift, relu))
                    i += consumed
                    continue
                pre = _train_bn_pre(module, mods, i, input) if isinstance(input, SparseConvTensor) else None
                if pre is not None:
                    # training: conv + BatchNorm1d (batch statistics) + ReLU as one autograd node over the libspx kernels
                    bn, relu, consumed = pre
                    input = module(input, train_bn=(bn, relu))
                    i += consumed
                    continue
                input = module(input)
                tail = _train_bn_tail(mods, i, input) if isinstance(input, SparseConvTensor) else None
                if tail is not None:
                    # training: BatchNorm1d (batch statistics) + ReLU as one libspx kernel pair instead of 3 + 3 launches
                    bn, relu, consumed = tail
                    input = input.replace_feature(F_.bn_relu_train(input.features, bn, relu, d_n=input.n_valid))
                    i += consumed
                    continue
            elif isinstance(input, SparseConvTensor):
                if input.n_valid is not None:
                    if _row_local(module):
                        # e.g. BatchNorm1d on its running statistics while the model trains (frozen BatchNorm, fine-tuning)
                        input = _on_live_rows(module, input)
                        i += 1
                        continue
                    raise RuntimeError("static-capacity tensors carry garbage rows beyond n_valid: only sparse modules, "
                                       "their fused BatchNorm/ReLU and row-local modules (eval BatchNorm1d, ReLU) may "
                                       "touch them")
                if input.indices.shape[0] != 0:
                    input = input.replace_feature(module(input.features))
            else:
                input = module(input)
            i += 1
        return input


def _row_local(module):
    """Modules whose output row depends on the same input row only (and on no batch statistic)."""
    return isinstance(module, (nn.ReLU, nn.Identity)) or (isinstance(module, nn.BatchNorm1d) and not module.training
                                                          and module.running_mean is not None)


def _on_live_rows(module, x):
    """A row-local torch module on a static-capacity tensor.  Rows beyond n_valid hold whatever the buffer held (possibly
    NaN) and so do the rows of the gradient that comes back: both are replaced by zeros around the module — a select, not
    a multiply — so that neither the module's output nor its parameter gradients (sums over rows) ever see them."""
    f = x.features
    live = (torch.arange(f.shape[0], device=f.device) < x.n_valid).unsqueeze(1)
    zero = torch.zeros((), dtype=f.dtype, device=f.device)
    y = module(torch.where(live, f, zero))
    return x.replace_feature(torch.where(live, y, zero))


_FUSED_BLOCK = os.environ.get("SPX_FUSED_BLOCK", "1") != "0"     # dev knob: conv and BN+ReLU as separate autograd nodes


def _train_bn_pre(conv, mods, i, x):
    """(bn, relu, modules consumed) when mods[i:] is a non-inverse SparseConvolution followed by a training-mode
    BatchNorm1d[, ReLU] that libspx covers — decided before the conv runs, from its out_channels — so that the three run
    as one autograd node (F_.sparse_conv_bn_relu)."""
    if not _FUSED_BLOCK or not is_sparse_conv(conv) or conv.inverse or i + 1 >= len(mods) or not x.features.is_cuda:
        return None
    bn = mods[i + 1]
    if not (isinstance(bn, nn.BatchNorm1d) and bn.training and bn.affine and bn.track_running_stats
            and F_.bn_momentum_ok(bn) and x.features.dtype == torch.float32 and ops.bn_relu_supported(conv.out_channels)
            and x.features.shape[0] > 1 and torch.is_grad_enabled()):
        return None
    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
    return bn, relu, (3 if relu else 2)


def _train_bn_tail(mods, i, x):
    """(bn, relu, modules consumed) when mods[i+1:] starts with a training-mode affine BatchNorm1d[, ReLU] that libspx's
    fused kernels cover; else None (the torch modules then run as usual)."""
    if i + 1 >= len(mods):
        return None
    bn = mods[i + 1]
    f = x.features
    if not F_.bn_train_fusable(bn, f):
        return None
    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
    return bn, relu, (3 if relu else 2)


def _fusable_tail(conv, mods, i):
    """(scale, shift, relu, modules consumed) when mods[i:] is SparseConvolution, BatchNorm1d(eval)[, ReLU] and no
    autograd graph is being recorded (the fused epilogue has no backward); else None."""
    if not is_sparse_conv(conv) or torch.is_grad_enabled() or i + 1 >= len(mods):
        return None
    bn = mods[i + 1]
    if not isinstance(bn, nn.BatchNorm1d) or bn.training or bn.running_mean is None:
        return None
    scale, shift, _ = F_.folded_bn(bn, conv.bias)
    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
    return scale, shift, relu, (3 if relu else 2)


def _triple(v, ndim=3):
    if isinstance(v, (list, tuple)):
        assert len(v) == ndim
        return [int(x) for x in v]
    return [int(v)] * ndim


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 algo=None, fp32_accum=None, name=None, **_unused):
        super().__init__()
        if ndim != 3:
            raise NotImplementedError("libspx implements 3-D sparse convolution only (the reference uses no other)")
        if groups != 1:
            raise NotImplementedError("groups != 1 is not used by the reference and not implemented")
        if transposed:
            raise NotImplementedError("SparseConvTranspose3d is not used by the reference and not implemented")
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.dilation = _triple(dilation)
        self.output_padding = _triple(output_padding)
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.subm, self.inverse, self.transposed = subm, inverse, transposed
        self.groups = groups
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # same fan-in (K * Cin) and bound as torch's conv default (kaiming_uniform(a=sqrt(5)))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1.0 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        s = "{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, padding={padding}"
        if self.subm:
            s += ", subm"
        if self.inverse:
            s += ", inverse"
        if self.indice_key is not None:
            s += ", indice_key={indice_key}"
        return s.format(**self.__dict__)

    def _rulebook(self, x):
        """Find (indice_key cache) or build the rulebook; returns (rulebook, out_indices, out_shape)."""
        cached = x.find_indice_pair(self.indice_key)
        if cached is not None:
            _prebuild.wait_ready(cached)        # a table queued on the index stream (spx/prebuild.py)
        if self.inverse:
            if cached is None:
                raise ValueError("SparseInverseConv3d needs the rulebook of the conv with indice_key=%r"
                                 % self.indice_key)
            if cached.n_out != x.indices.shape[0]:
                raise ValueError("inverse conv input does not match the cached forward conv output")
            return cached
        if cached is not None:
            if self.subm and (cached.n_in != x.indices.shape[0] or cached.ksize != self.kernel_size):
                raise ValueError("indice_key %r was built for a different tensor / kernel" % self.indice_key)
            return cached
        if self.subm:
            # unique: spconv's contract for SparseConvTensor.indices (one row per active cell)
            rb = ops.subm_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.dilation,
                                   d_n=x.n_valid, unique=True)
        elif x.n_valid is not None:     # static-capacity mode: no host sync, rows at capacity
            cap = (x.static_caps or {}).get(self.indice_key, None)
            rb = ops.conv_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride,
                                   self.padding, self.dilation, d_n_in=x.n_valid, cap=cap, sync=False)
        else:
            rb = ops.conv_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride,
                                   self.padding, self.dilation)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = rb
        return rb

    def forward(self, x, fused=None, train_bn=None):
        assert isinstance(x, SparseConvTensor)
        feats = x.features
        if feats.shape[1] != self.in_channels:
            raise ValueError("channel size mismatch: got %d, conv expects %d" % (feats.shape[1], self.in_channels))
        rb = self._rulebook(x)
        if train_bn is not None and not self.inverse and rb.n_out > 1 and rb.n_in > 0:
            out_feats = F_.sparse_conv_bn_relu(feats, self.weight, self.bias, rb, train_bn[0], train_bn[1])
        elif train_bn is not None:
            out_feats = F_.sparse_conv(feats, self.weight, self.bias, rb, inverse=self.inverse)
            out_feats = F_.bn_act(out_feats, train_bn[0], train_bn[1], d_n=(rb.d_n_in if self.inverse else rb.d_n_out))
        elif fused is not None:
            scale, shift, relu = fused      # shift already contains the conv bias
            out_feats = F_.sparse_conv(feats, self.weight, None, rb, inverse=self.inverse, scale=scale, shift=shift,
                                       relu=relu)
        else:
            out_feats = F_.sparse_conv(feats, self.weight, self.bias, rb, inverse=self.inverse)
        if self.inverse:
            out = SparseConvTensor(out_feats, self._inverse_indices(x, rb), rb.in_shape,
                                   x.batch_size, x.grid, x.voxel_num, x.indice_dict, x.benchmark, n_valid=rb.d_n_in,
                                   static_caps=x.static_caps)
        elif self.subm:
            out = x.replace_feature(out_feats)
        else:
            out = SparseConvTensor(out_feats, rb.out_indices, rb.out_shape, x.batch_size, x.grid, x.voxel_num,
                                   x.indice_dict, x.benchmark, n_valid=rb.d_n_out, static_caps=x.static_caps)
        return out

    @staticmethod
    def _inverse_indices(x, rb):
        idx = getattr(rb, "in_indices", None)
        if idx is None:
            raise ValueError("rulebook does not carry the forward conv's input indices")
        return idx


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None, **kw):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, subm=True,
                         indice_key=indice_key, algo=algo, **kw)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None, **kw):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key, algo=algo, **kw)

    def _rulebook(self, x):
        rb = super()._rulebook(x)
        if getattr(rb, "in_indices", None) is None:
            rb.in_indices = x.indices  # kept for SparseInverseConv3d with the same indice_key
        return rb


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True, algo=None, fp32_accum=None,
                 name=None, **kw):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key,
                         algo=algo, **kw)


class ToDense(SparseModule):
    def forward(self, x):
        return x.dense()
