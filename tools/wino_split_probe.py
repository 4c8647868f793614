"""Dev tool (GPU box): does running the BEV convolutions as two independent half-batch chains on two streams beat one
full-batch chain?  (Inference only: BatchNorm folded, no batch statistics between the layers.)  A full-batch launch of the
256-channel maps is 1 100 workgroups on 512 slots = 2.15 rounds -> 3; two chains fill each other's last rounds."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
from spx import ops  # noqa: E402

dev = torch.device("cuda:0")


def chain(x, us, scale, shift):
    for u in us:
        x = ops.conv2d_wino(x, u, scale.numel(), scale=scale, shift=shift, relu=True)
    return x


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (n, c, h, w) in ((4, 128, 200, 176), (4, 256, 100, 88), (2, 128, 188, 188), (2, 256, 94, 94)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((n, c, h, w), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn((c, c, 3, 3), generator=g) / (3 * c ** 0.5)).to(dev) for _ in range(5)]
    us = [ops.wino_weight(wt) for wt in ws]
    scale = (torch.rand(c, generator=g) + 0.5).to(dev)
    shift = torch.randn(c, generator=g).to(dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    xa, xb = x[: n // 2], x[n // 2:]
    ss = [torch.cuda.Stream() for _ in range(n)]

    def per_frame():
        cur = torch.cuda.current_stream()
        outs = []
        for i in range(n):
            ss[i].wait_stream(cur)
            with torch.cuda.stream(ss[i]):
                outs.append(chain(x[i:i + 1], us, scale, shift))
        for i in range(n):
            cur.wait_stream(ss[i])
        return outs

    def full():
        return chain(x, us, scale, shift)

    def split():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ya = chain(xa, us, scale, shift)
        with torch.cuda.stream(s2):
            yb = chain(xb, us, scale, shift)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        return ya, yb

    ya, yb = split()
    yf = full()
    torch.cuda.synchronize()
    same = torch.equal(yf[: n // 2], ya) and torch.equal(yf[n // 2:], yb)
    # graphs: launch overhead out of the picture
    gf, gs, gp = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf):
        full()
    with torch.cuda.graph(gs):
        split()
    with torch.cuda.graph(gp):
        per_frame()
    print("%dx%dx%dx%d  five convs: full batch %7.1f us   two half-batch chains %7.1f us   (graphs: %7.1f / %7.1f; one chain per frame %7.1f)  bit-identical: %s" % (
        n, c, h, w, timeit(full), timeit(split), timeit(gf.replay), timeit(gs.replay), timeit(gp.replay), same), flush=True)
