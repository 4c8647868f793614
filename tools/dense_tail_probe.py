"""Dev probe (GPU): alternatives for pieces of the dense BEV tail at the bench shapes (cfg 2, batch 4, NHWC, fp32).
python tools/dense_tail_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last


def bench(name, fn, iters=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-58s %9.1f us" % (name, e0.elapsed_time(e1) / iters * 1e3), flush=True)


def fb(mod_fn, x, g):
    def run():
        x.grad = None
        y = mod_fn(x)
        y.backward(g)
    return run


def deconv_gemm(x, w, s):
    """ConvTranspose2d with kernel == stride == s, as one GEMM + pixel shuffle.  x NCHW logical / channels_last memory."""
    B, Ci, H, W = x.shape
    Co = w.shape[1]
    xm = x.permute(0, 2, 3, 1).reshape(B * H * W, Ci)
    wm = w.permute(0, 2, 3, 1).reshape(Ci, s * s * Co)              # [Ci, (i, j, co)]
    y = (xm @ wm).view(B, H, W, s, s, Co).permute(0, 1, 3, 2, 4, 5).reshape(B, H * s, W * s, Co)
    return y.permute(0, 3, 1, 2)


def main():
    torch.manual_seed(0)
    # A: deblock 1
    x = torch.randn(4, 256, 100, 88, device=dev).to(memory_format=CL).requires_grad_(True)
    up = nn.ConvTranspose2d(256, 256, 2, stride=2, bias=False).to(dev).to(memory_format=CL)
    g = torch.randn(4, 256, 200, 176, device=dev).to(memory_format=CL)
    with torch.no_grad():
        ref = up(x)
        alt = deconv_gemm(x, up.weight, 2)
        print("deconv2x2 gemm vs ConvTranspose2d max|diff| %.3e (max|ref| %.3f) alt strides %s" % (
            (ref - alt).abs().max().item(), ref.abs().max().item(), alt.stride()))
        bench("A deconv2x2 ConvTranspose2d fwd", lambda: up(x))
        bench("A deconv2x2 gemm+shuffle    fwd", lambda: deconv_gemm(x, up.weight, 2))
    bench("A deconv2x2 ConvTranspose2d fwd+bwd", fb(up, x, g))
    bench("A deconv2x2 gemm+shuffle    fwd+bwd", fb(lambda t: deconv_gemm(t, up.weight, 2), x, g))

    # B: deblock 0
    x0 = torch.randn(4, 128, 200, 176, device=dev).to(memory_format=CL).requires_grad_(True)
    up0 = nn.ConvTranspose2d(128, 256, 1, stride=1, bias=False).to(dev).to(memory_format=CL)
    with torch.no_grad():
        bench("B deconv1x1 ConvTranspose2d fwd", lambda: up0(x0))
        bench("B deconv1x1 gemm            fwd", lambda: deconv_gemm(x0, up0.weight, 1))
        bench("B deconv1x1 F.conv2d(w^T)   fwd", lambda: F.conv2d(x0, up0.weight.permute(1, 0, 2, 3)))
    bench("B deconv1x1 ConvTranspose2d fwd+bwd", fb(up0, x0, g))
    bench("B deconv1x1 gemm            fwd+bwd", fb(lambda t: deconv_gemm(t, up0.weight, 1), x0, g))
    bench("B deconv1x1 F.conv2d(w^T)   fwd+bwd", fb(lambda t: F.conv2d(t, up0.weight.permute(1, 0, 2, 3)), x0, g))

    # cat
    a = torch.randn(4, 256, 200, 176, device=dev).to(memory_format=CL)
    b = torch.randn(4, 256, 200, 176, device=dev).to(memory_format=CL)
    bench("cat([a, b], 1) channels_last", lambda: torch.cat([a, b], 1))

    # C: heads
    xs = torch.randn(4, 512, 200, 176, device=dev).to(memory_format=CL).requires_grad_(True)
    heads = [nn.Conv2d(512, c, 1).to(dev).to(memory_format=CL) for c in (18, 42, 12)]
    gs = [torch.randn(4, c, 200, 176, device=dev).to(memory_format=CL) for c in (18, 42, 12)]
    gcat = torch.cat(gs, 1).contiguous(memory_format=CL)

    def sep():
        xs.grad = None
        for h in heads:
            h.weight.grad = None
        ys = [h(xs) for h in heads]
        torch.autograd.backward(ys, gs)

    def fused():
        xs.grad = None
        for h in heads:
            h.weight.grad = None
        w = torch.cat([h.weight for h in heads], 0)
        bb = torch.cat([h.bias for h in heads], 0)
        y = F.conv2d(xs, w, bb)
        y.backward(gcat)

    def fused_mm():
        xs.grad = None
        for h in heads:
            h.weight.grad = None
        w = torch.cat([h.weight.view(h.weight.shape[0], -1) for h in heads], 0)
        bb = torch.cat([h.bias for h in heads], 0)
        y = F.linear(xs.permute(0, 2, 3, 1), w, bb)
        y.backward(gcat.permute(0, 2, 3, 1))

    with torch.no_grad():
        bench("C heads 3 separate 1x1 fwd", lambda: [h(xs) for h in heads])
        w = torch.cat([h.weight for h in heads], 0)
        bench("C heads fused conv 72  fwd", lambda: F.conv2d(xs, w))
        bench("C heads fused linear   fwd", lambda: F.linear(xs.permute(0, 2, 3, 1), w.view(72, 512)))
    w72 = torch.cat([h.weight for h in heads], 0).detach()
    b72 = torch.cat([h.bias for h in heads], 0).detach()
    xd = xs.detach()

    def lin_fwd_conv_bwd():
        y = F.linear(xd.permute(0, 2, 3, 1), w72.view(72, 512), b72)
        return torch.ops.aten.convolution_backward(gcat, xd, w72, [72], [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                   [True, True, True]), y

    def lin_fwd_mixed_bwd():
        y = F.linear(xd.permute(0, 2, 3, 1), w72.view(72, 512), b72)
        g2 = gcat.permute(0, 2, 3, 1).reshape(-1, 72)
        dx = g2 @ w72.view(72, 512)
        rest = torch.ops.aten.convolution_backward(gcat, xd, w72, [72], [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                   [False, True, True])
        return dx, rest, y

    def lin_fwd_mixed_bwd2():
        y = F.linear(xd.permute(0, 2, 3, 1), w72.view(72, 512), b72)
        g2 = gcat.permute(0, 2, 3, 1).reshape(-1, 72)
        dw = g2.t() @ xd.permute(0, 2, 3, 1).reshape(-1, 512)
        rest = torch.ops.aten.convolution_backward(gcat, xd, w72, [72], [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                   [True, False, True])
        return dw, rest, y

    bench("C heads linear fwd + conv bwd (dx, dw, db)", lin_fwd_conv_bwd)
    bench("C heads linear fwd + mm dx + conv (dw, db)", lin_fwd_mixed_bwd)
    bench("C heads linear fwd + mm dw + conv (dx, db)", lin_fwd_mixed_bwd2)
    bench("C heads 3 separate 1x1 fwd+bwd", sep)
    bench("C heads fused conv 72  fwd+bwd", fused)
    bench("C heads fused linear   fwd+bwd", fused_mm)

    # D: ZeroPad2d + conv(pad 0)  vs  conv(pad 1)
    xi = torch.randn(4, 256, 200, 176, device=dev).to(memory_format=CL).requires_grad_(True)
    c0 = nn.Conv2d(256, 128, 3, stride=1, padding=0, bias=False).to(dev).to(memory_format=CL)
    g0 = torch.randn(4, 128, 200, 176, device=dev).to(memory_format=CL)
    bench("D s1 ZeroPad2d+conv(p0) fwd+bwd", fb(lambda t: c0(F.pad(t, (1, 1, 1, 1))), xi, g0))
    bench("D s1 conv(p1)           fwd+bwd", fb(lambda t: F.conv2d(t, c0.weight, None, 1, 1), xi, g0))
    xj = torch.randn(4, 128, 200, 176, device=dev).to(memory_format=CL).requires_grad_(True)
    c1 = nn.Conv2d(128, 256, 3, stride=2, padding=0, bias=False).to(dev).to(memory_format=CL)
    g1 = torch.randn(4, 256, 100, 88, device=dev).to(memory_format=CL)
    bench("D s2 ZeroPad2d+conv(p0) fwd+bwd", fb(lambda t: c1(F.pad(t, (1, 1, 1, 1))), xj, g1))
    bench("D s2 conv(p1)           fwd+bwd", fb(lambda t: F.conv2d(t, c1.weight, None, 2, 1), xj, g1))

    # E: BN2d + ReLU (train)  vs  spx fused bn_relu on the [B*H*W, C] view
    from spx.functional import bn_relu_train
    for C, H, W in ((128, 200, 176), (256, 100, 88), (256, 200, 176)):
        xb = torch.randn(4, C, H, W, device=dev).to(memory_format=CL).requires_grad_(True)
        gb = torch.randn(4, C, H, W, device=dev).to(memory_format=CL)
        bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(dev).train()
        bn1 = nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()

        def torch_bn(t):
            return F.relu(bn(t))

        def spx_bn(t):
            y = bn_relu_train(t.permute(0, 2, 3, 1).reshape(-1, C), bn1, True)
            return y.view(4, H, W, C).permute(0, 3, 1, 2)

        with torch.no_grad():
            pass
        ya, yb = torch_bn(xb), spx_bn(xb)
        print("E C=%d max|diff| %.3e" % (C, (ya - yb).abs().max().item()))
        bench("E BN2d+ReLU torch C=%d %dx%d fwd+bwd" % (C, H, W), fb(torch_bn, xb, gb))
        bench("E BN+ReLU   spx   C=%d %dx%d fwd+bwd" % (C, H, W), fb(spx_bn, xb, gb))


if __name__ == "__main__":
    main()
