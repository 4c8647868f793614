"""Time csrc/wino_conv2d.hip against the vendor convolution on the BEV shapes of BASELINE configs[1] (KITTI, batch 4).
usage: python tools/wino_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsm-det-pointcloud-_amd"))


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-db", action="store_true")
    args = ap.parse_args()
    if not args.no_db:
        from pcdet_amd.utils.miopen_db import use_tuned_db
        use_tuned_db()
    from spx import ops
    for (n, c, h, w) in [(4, 128, 200, 176), (4, 256, 100, 88), (1, 128, 200, 176), (1, 256, 100, 88)]:
        x = torch.randn((n, c, h, w), device="cuda").contiguous(memory_format=torch.channels_last)
        wt = (torch.randn((c, c, 3, 3), device="cuda") / (3 * c ** 0.5)).contiguous(memory_format=torch.channels_last)
        u = ops.wino_weight(wt)
        out = torch.empty_like(x)
        t_w = timeit(lambda: ops.conv2d_wino(x, u, c, out=out), args.iters)
        t_v = timeit(lambda: F.conv2d(x, wt, padding=1), args.iters)
        t_u = timeit(lambda: ops.wino_weight(wt), args.iters)
        dy = torch.randn_like(x)
        t_gw = timeit(lambda: ops.conv2d_wino_wgrad(x, dy, wt), args.iters)
        t_gv = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                                   (False, True, False)), args.iters)
        gf = 2.0 * 9 * c * c * n * h * w / 1e9
        err = float((out - F.conv2d(x, wt, padding=1)).abs().max())
        print("%dx%dx%dx%d: wino %.1f us (%.1f TF/s direct-equivalent, %.1f TF/s executed)  vendor %.1f us (%.1f TF/s)  "
              "weight transform %.1f us  max|diff| %.2e" % (n, c, h, w, t_w, gf / t_w * 1e3, gf / 2.25 / t_w * 1e3, t_v,
                                                             gf / t_v * 1e3, t_u, err), flush=True)
        print("      weight gradient: wino %.1f us (%.1f TF/s direct-equivalent)  vendor %.1f us (%.1f TF/s)"
              % (t_gw, gf / t_gw * 1e3, t_gv, gf / t_gv * 1e3), flush=True)


if __name__ == "__main__":
    main()
