"""Dev tool (GPU box): the BatchNorm kernels alone at the BEV shapes of BASELINE configs[1] (rows x channels of the
channels-last maps) and at a sparse-layer shape — run under `rocprofv3 --kernel-trace --stats` for per-kernel durations.

python tools/bnbench.py [--iters 20]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from spx import ops
    dev = torch.device("cuda:0")
    for n, c in ((4 * 200 * 176, 128), (4 * 100 * 88, 256), (4 * 200 * 176, 256), (82202, 64)):
        x = torch.randn(n, c, device=dev)
        dy = torch.randn(n, c, device=dev)
        g = torch.rand(c, device=dev) + 0.5
        b = torch.randn(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        y, mean, invstd = ops.bn_relu_fwd(x, g, b, rm, rv, 0.01, 1e-3, True)
        t = []
        for what in ("fwd", "bwd"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f = (lambda: ops.bn_relu_fwd(x, g, b, rm, rv, 0.01, 1e-3, True)) if what == "fwd" else \
                (lambda: ops.bn_relu_bwd(x, dy, g, b, mean, invstd, True))
            for _ in range(3):
                f()
            e0.record()
            for _ in range(args.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / args.iters * 1e3)
        mb = n * c * 4 / 1e6
        print("rows %7d x %3d ch (%6.1f MB): fwd %7.1f us (3 passes: %5.2f TB/s)  bwd %7.1f us (5 passes: %5.2f TB/s)" % (
            n, c, mb, t[0], 3 * mb / t[0], t[1], 5 * mb / t[1]), flush=True)


if __name__ == "__main__":
    main()
