"""Winograd F(2x2, 3x3) dense convolution (csrc/wino_conv2d.hip) against a float64 direct convolution on the CPU.

Bar: |err| <= 2e-5 * max(1, max|ref|), the same as the sparse fp32-MFMA convolutions: arithmetic is fp32 throughout, the
transform only re-associates the sum (measured ~1e-6 * scale at 128..256 input channels).  torch's fp32 conv2d on the same
GPU (MIOpen, direct form) is checked against the same reference beside it, so the two error levels can be compared.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _ref(x, w, scale=None, shift=None, relu=False):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    if scale is not None:
        y = y * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
    if relu:
        y = torch.relu(y)
    return y


def _check(y, ref, tol=TOL):
    ref = ref.numpy()
    err = float(np.abs(y.double().cpu().numpy() - ref).max())
    scale = max(1.0, float(np.abs(ref).max()))
    assert err <= tol * scale, "max|err| %.3e > %.1e * %.3g" % (err, tol, scale)
    return err / scale


@pytest.mark.parametrize("n,h,w,cin,cout", [(1, 8, 8, 32, 128), (1, 7, 9, 64, 128), (2, 20, 18, 96, 256), (3, 5, 3, 32, 128),
                                             (1, 1, 1, 32, 128), (2, 33, 17, 128, 128), (1, 26, 22, 256, 256)])
def test_forward_matches_direct_conv(n, h, w, cin, cout):
    from spx import ops
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    x = torch.randn((n, cin, h, w), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)).cuda()
    y = ops.conv2d_wino(x, ops.wino_weight(wt), cout)
    assert y.shape == (n, cout, h, w) and y.is_contiguous(memory_format=torch.channels_last)
    ref = _ref(x, wt)
    e_w = _check(y, ref)
    e_t = _check(F.conv2d(x, wt, padding=1), ref)
    print("rel err wino %.2e  vendor direct %.2e" % (e_w, e_t))


def test_channels_last_weight_and_epilogue():
    from spx import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 32, 10, 12), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((128, 32, 3, 3), generator=g) / 12).cuda().contiguous(memory_format=torch.channels_last)
    sc = (torch.rand((128,), generator=g) + 0.5).cuda()
    sh = torch.randn((128,), generator=g).cuda()
    y = ops.conv2d_wino(x, ops.wino_weight(wt), 128, scale=sc, shift=sh, relu=True)
    _check(y, _ref(x, wt, sc, sh, True))


def test_row_strided_input_and_output():
    """x is a channel slice of a wider map, y is written into a channel slice of a wider map (the concatenated BEV map)."""
    from spx import ops
    g = torch.Generator().manual_seed(6)
    big = torch.randn((2, 96, 9, 11), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    x = big[:, 32:64]
    wt = (torch.randn((128, 32, 3, 3), generator=g) / 12).cuda()
    out_big = torch.full((2, 384, 9, 11), 7.0).cuda().contiguous(memory_format=torch.channels_last)
    ops.conv2d_wino(x, ops.wino_weight(wt), 128, out=out_big[:, 128:256])
    _check(out_big[:, 128:256], _ref(x, wt))
    assert float(out_big[:, :128].min()) == 7.0 and float(out_big[:, 256:].max()) == 7.0


def test_data_gradient_image():
    """conv2d_wino(dy, wino_weight(w, flip=True)) is the data gradient of conv2d(x, w, padding=1)."""
    from spx import ops
    g = torch.Generator().manual_seed(7)
    cin, cout = 128, 32
    wt = (torch.randn((cout, cin, 3, 3), generator=g) / 12).cuda()
    dy = torch.randn((2, cout, 12, 14), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    dx = ops.conv2d_wino(dy, ops.wino_weight(wt, flip=True), cin)
    x = torch.zeros((2, cin, 12, 14), dtype=torch.float64, requires_grad=True)
    F.conv2d(x, wt.double().cpu(), padding=1).backward(dy.double().cpu())
    _check(dx, x.grad)


def test_kitti_bev_shape_linearity_and_vendor_agreement():
    """Full BEV size of BASELINE configs[1] (4 x 128 x 200 x 176): agreement with the vendor convolution and linearity."""
    from spx import ops
    g = torch.Generator().manual_seed(8)
    x1 = torch.randn((4, 128, 200, 176), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    x2 = torch.randn((4, 128, 200, 176), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((128, 128, 3, 3), generator=g) / np.sqrt(9 * 128)).cuda()
    u = ops.wino_weight(wt)
    y1, y2 = ops.conv2d_wino(x1, u, 128), ops.conv2d_wino(x2, u, 128)
    y12 = ops.conv2d_wino(x1 + 2 * x2, u, 128)
    scale = float(y12.abs().max())
    assert float((y12 - (y1 + 2 * y2)).abs().max()) <= TOL * scale
    assert float((y1 - F.conv2d(x1, wt, padding=1)).abs().max()) <= TOL * scale


def test_rejects_unsupported_channels():
    from spx import ops
    wt = torch.zeros((64, 32, 3, 3)).cuda()
    assert not ops.wino_ok(32, 64) and ops.wino_ok(32, 128) and not ops.wino_ok(8, 128)
    with pytest.raises(RuntimeError):
        ops.wino_weight(wt)


@pytest.mark.parametrize("n,h,w,cin,cout", [(1, 16, 16, 128, 128), (2, 15, 19, 128, 128), (3, 33, 17, 128, 256),
                                             (1, 26, 22, 256, 128), (2, 9, 30, 256, 256)])
def test_weight_gradient_matches_float64(n, h, w, cin, cout):
    """Winograd-domain weight gradient (csrc/wino_wgrad.hip) against conv2d autograd in float64; the vendor's fp32 weight
    gradient is measured against the same reference beside it."""
    from spx import ops
    g = torch.Generator().manual_seed(n * 100 + h + w)
    x = torch.randn((n, cin, h, w), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    dy = torch.randn((n, cout, h, w), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wt = torch.zeros((cout, cin, 3, 3), device="cuda")
    dw = ops.conv2d_wino_wgrad(x, dy, wt)
    wr = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wr, padding=1).backward(dy.double().cpu())
    e_w = _check(dw, wr.grad.detach())
    dv = torch.ops.aten.convolution_backward(dy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    e_v = _check(dv, wr.grad.detach(), tol=1e-4)
    print("rel err wino wgrad %.2e  vendor wgrad %.2e" % (e_w, e_v))
    # twice the same bits (fixed summation order, no atomics)
    assert torch.equal(dw, ops.conv2d_wino_wgrad(x, dy, wt))


def test_weight_gradient_channels_last_weight_and_slices():
    from spx import ops
    g = torch.Generator().manual_seed(77)
    big = torch.randn((2, 384, 12, 20), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    x = big[:, 128:256]
    dyb = torch.randn((2, 256, 12, 20), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    dy = dyb[:, 128:]
    like = torch.zeros((128, 128, 3, 3), device="cuda").contiguous(memory_format=torch.channels_last)
    dw = ops.conv2d_wino_wgrad(x, dy, like)
    assert dw.stride() == like.stride()
    wr = torch.zeros((128, 128, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wr, padding=1).backward(dy.double().cpu())
    _check(dw, wr.grad.detach())
    assert not ops.wino_wgrad_ok(128, 128, 13) and ops.wino_wgrad_ok(128, 128, 15) and not ops.wino_wgrad_ok(64, 128, 100)


def test_autograd_matches_vendor_conv():
    """_WinoConv2dFn: output, data gradient (Winograd kernel with the flipped image) and weight gradient (vendor wrw)
    against torch's conv2d autograd in float64."""
    from spx.functional import wino_conv2d
    g = torch.Generator().manual_seed(9)
    x = torch.randn((2, 128, 14, 18), generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn((128, 128, 3, 3), generator=g) / 34).cuda().requires_grad_(True)
    dy = torch.randn((2, 128, 14, 18), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    y = wino_conv2d(x, w)
    y.backward(dy)
    xr = x.detach().double().cpu().requires_grad_(True)
    wr = w.detach().double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    yr.backward(dy.double().cpu())
    _check(y.detach(), yr.detach())
    _check(x.grad, xr.grad)
    _check(w.grad, wr.grad, tol=5e-5)


def _bev(train):
    from pcdet_amd.config import AttrDict as EasyDict
    from pcdet_amd.models.backbones_2d.base_bev_backbone import BaseBEVBackbone
    cfg = EasyDict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[128, 256], UPSAMPLE_STRIDES=[1, 2],
                   NUM_UPSAMPLE_FILTERS=[256, 256])
    torch.manual_seed(3)
    m = BaseBEVBackbone(cfg, 64).cuda().to(memory_format=torch.channels_last)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
            # pre-activations well away from 0: the ReLU that follows has a kink there, and an element within fp32 round-off
            # of it makes the gradient of a whole channel jump (seen: one element at 7e-7 flipping between two runs of the
            # SAME path) — nothing to do with which convolution kernel ran
            mod.weight.data.uniform_(0.1, 0.3)
            mod.bias.data.normal_(2.5, 0.1)
    return m.train(train)


@pytest.mark.parametrize("train", [False, True])
def test_bev_backbone_with_and_without_winograd(train):
    """BaseBEVBackbone forward (and, training, every parameter gradient) with the Winograd rewrite on and off, both against
    the same modules run in float64 on the CPU.  Train-mode BatchNorm over these small maps makes the weight gradients of
    the convolutions before it differences of nearly cancelling terms, so fp32 round-off anywhere upstream is amplified:
    the bar for the Winograd path is the error the vendor path itself shows against float64 (x4, floor 1e-4)."""
    import copy
    import pcdet_amd.models.backbones_2d.base_bev_backbone as bb
    m = _bev(train)
    x = torch.randn((2, 64, 24, 20), generator=torch.Generator().manual_seed(4)).cuda().contiguous(memory_format=torch.channels_last)
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def run(model, inp, wt):
        for p in model.parameters():
            p.grad = None
        if train:
            y = model({'spatial_features': inp})['spatial_features_2d']
            (y * wt.to(y.dtype).to(y.device).view_as(y)).sum().backward()
            return y.detach().double().cpu(), [p.grad.detach().double().cpu() for p in model.parameters()]
        with torch.no_grad():
            y = model({'spatial_features': inp})['spatial_features_2d']
        return y.double().cpu(), []

    m64 = copy.deepcopy(m).double().cpu().train(train)
    wt = None
    with torch.no_grad():
        wt = torch.linspace(0.5, 1.5, 2 * 512 * 24 * 20, dtype=torch.float64)
    y_ref, g_ref = run(m64, x.double().cpu(), wt)
    errs = {}
    for flag in (False, True):
        bb._WINO = flag
        try:
            m.load_state_dict(state)
            y, g = run(m, x, wt)
        finally:
            bb._WINO = True
        e_y = float((y - y_ref).abs().max() / y_ref.abs().max())
        per = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-3)) for a, b in zip(g, g_ref)]
        e_g = max(per or [0.0])
        if per:
            names = [k for k, _ in m.named_parameters()]
            print("winograd=%s worst parameter gradients:" % flag,
                  sorted(zip(per, names, [tuple(p.shape) for p in m.parameters()]), reverse=True)[:4])
        errs[flag] = (e_y, e_g)
    print("vs float64: vendor path output %.2e grads %.2e | winograd path output %.2e grads %.2e"
          % (errs[False] + errs[True]))
    assert errs[True][0] <= max(4 * errs[False][0], 1e-5)
    assert errs[True][1] <= max(4 * errs[False][1], 1e-4)


def test_autograd_rectangular_channels_fall_back_per_direction():
    """64 -> 128 channels: forward on the Winograd kernel (Cin % 16, Cout % 128), data gradient (needs Cin % 128) and weight
    gradient (needs both % 128) through the vendor library — every direction still matches float64."""
    from spx.functional import wino_conv2d
    g = torch.Generator().manual_seed(10)
    x = torch.randn((2, 64, 10, 16), generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn((128, 64, 3, 3), generator=g) / 24).cuda().requires_grad_(True)
    dy = torch.randn((2, 128, 10, 16), generator=g).cuda().contiguous(memory_format=torch.channels_last)
    wino_conv2d(x, w).backward(dy)
    xr = x.detach().double().cpu().requires_grad_(True)
    wr = w.detach().double().cpu().requires_grad_(True)
    F.conv2d(xr, wr, padding=1).backward(dy.double().cpu())
    _check(x.grad, xr.grad)
    _check(w.grad, wr.grad, tol=5e-5)


def test_conv_bn_relu_node_takes_statistics_in_the_conv_epilogue():
    """_WinoConvBNReLUFn: conv + training-mode BatchNorm + ReLU as one node, the BatchNorm sums taken by the convolution
    kernel while it stores its output.  Output, running statistics and all four gradients against the torch modules in
    float64 (BatchNorm parameters chosen so that no ReLU input sits at the kink)."""
    from spx.functional import wino_conv_bn_relu_train
    g = torch.Generator().manual_seed(11)
    conv = torch.nn.Conv2d(128, 128, 3, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(128, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / 34)
        bn.weight.uniform_(0.1, 0.3, generator=g)
        bn.bias.normal_(2.5, 0.1, generator=g)
        bn.running_mean.normal_(0, 0.1, generator=g)
        bn.running_var.uniform_(0.5, 1.5, generator=g)
    import copy
    conv64, bn64 = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    conv, bn = conv.cuda().to(memory_format=torch.channels_last), bn.cuda()
    x = torch.randn((3, 128, 21, 26), generator=g)          # odd height: partly filled tiles in the sums
    wt = torch.randn((3, 128, 21, 26), generator=g)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = wino_conv_bn_relu_train(xg, conv, bn)
    (y * wt.cuda()).sum().backward()
    x64 = x.double().requires_grad_(True)
    y64 = torch.relu(bn64(conv64(x64)))
    (y64 * wt.double()).sum().backward()
    _check(y.detach(), y64.detach())
    _check(bn.running_mean, bn64.running_mean)
    _check(bn.running_var, bn64.running_var)
    assert int(bn.num_batches_tracked) == 1
    _check(xg.grad, x64.grad, tol=1e-4)
    _check(conv.weight.grad, conv64.weight.grad, tol=1e-4)
    _check(bn.weight.grad, bn64.weight.grad, tol=1e-4)
    _check(bn.bias.grad, bn64.bias.grad, tol=1e-4)


# ------------------------------------------------------------------------------------------ full BASELINE sizes, float64 on the GPU

def _taps64(x):
    """x [N, C, H, W] -> the nine zero-padded shifted views of x as float64 channels-last rows [9][N*H*W, C] (on the GPU)."""
    n, c, h, w = x.shape
    xp = F.pad(x.double().permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))               # [N, H+2, W+2, C]
    return [xp[:, a:a + h, b:b + w, :].reshape(n * h * w, c) for a in range(3) for b in range(3)]


def _conv64(x, wt):
    """float64 conv2d(x, wt, padding=1) as nine GEMMs on the GPU -> [N*H*W, Cout] rows."""
    w64 = wt.double()
    y = None
    for t, rows in enumerate(_taps64(x)):
        a, b = divmod(t, 3)
        part = rows @ w64[:, :, a, b].t()
        y = part if y is None else y + part
    return y


def _rows(t):
    n, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(n * h * w, c)


def _bar(got, ref, tol):
    err = float((got.double() - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert err <= tol * scale, "max|err| %.3e > %.1e * %.3g" % (err, tol, scale)
    return err / scale


@pytest.mark.parametrize("n,c,h,w", [(4, 128, 200, 176), (4, 256, 100, 88), (2, 128, 188, 188), (2, 256, 94, 94),
                                      (3, 128, 101, 77)])
def test_full_size_forward_dgrad_wgrad_and_bn_sums_against_float64(n, c, h, w):
    """The BEV layers of BASELINE configs[1] (KITTI: 4 x 128 x 200 x 176, 4 x 256 x 100 x 88) and configs[2] (Waymo:
    2 x 128 x 188 x 188, 2 x 256 x 94 x 94) at FULL size, plus an odd map (the last XCD range of tile blocks and the last
    weight-gradient split partly empty): forward, data gradient, weight gradient (its split ranges only exist at these
    sizes) and the BatchNorm sums of the forward's epilogue, each against float64 computed on the GPU as nine GEMMs.
    Reference: pcdet/models/backbones_2d/base_bev_backbone.py:38-49."""
    from spx import ops
    g = torch.Generator().manual_seed(n * 1000 + h)
    dev = torch.device("cuda:0")
    x = torch.randn((n, c, h, w), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((n, c, h, w), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((c, c, 3, 3), generator=g) / np.sqrt(9 * c)).to(dev)
    # forward + epilogue sums
    y, part = ops.conv2d_wino(x, ops.wino_weight(wt), c, stats=True)
    ref = _conv64(x, wt)
    e_f = _bar(_rows(y), ref, TOL)
    s = part.double().sum(0)
    assert float((s[0] - ref.sum(0)).abs().max()) <= 2e-5 * float(ref.abs().sum(0).max())
    assert float((s[1] - (ref * ref).sum(0)).abs().max()) <= 2e-5 * float((ref * ref).sum(0).max())
    del ref
    # data gradient = conv of dy with the rotated, transposed filter
    dx = ops.conv2d_wino(dy, ops.wino_weight(wt, flip=True), c)
    wflip = wt.flip(2, 3).transpose(0, 1).contiguous()
    e_d = _bar(_rows(dx), _conv64(dy, wflip), TOL)
    # weight gradient: dw[co, ci, a, b] = sum_pixels dy[p, co] * x[p shifted by (a, b), ci]
    dw = ops.conv2d_wino_wgrad(x, dy, wt)
    dy64 = _rows(dy).double()
    want = torch.empty((c, c, 3, 3), dtype=torch.float64, device=dev)
    for t, rows in enumerate(_taps64(x)):
        a, b = divmod(t, 3)
        want[:, :, a, b] = dy64.t() @ rows
    e_w = _bar(dw, want, 5e-5)
    assert torch.equal(dw, ops.conv2d_wino_wgrad(x, dy, wt))             # fixed summation order
    print("rel err fwd %.2e dgrad %.2e wgrad %.2e" % (e_f, e_d, e_w))
