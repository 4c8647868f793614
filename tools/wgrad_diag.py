"""Dev tool (GPU box): where do the cycles of k_wgrad_mfma go?  Builds a DIAGNOSTIC copy of libspx with per-block cycle
stamps (csrc/conv_wgrad.hip, -DSPX_WG_DIAG), runs one layer of the cfg-2 batch and prints, per block: lifetime, cycles in
the K-step loops, steps, in-kernel clock.  The shipped library carries no stamps.

python tools/wgrad_diag.py [--layer conv3.1.0] [--burst 4]"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="conv3.1.0")
    ap.add_argument("--burst", type=int, default=4)
    args = ap.parse_args()
    csrc = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "csrc")
    out = "/tmp/spx_diag"
    os.makedirs(out, exist_ok=True)
    srcs = [f for f in os.listdir(csrc) if f.endswith(".hip")]
    objs = []
    for f in srcs:
        o = os.path.join(out, f[:-4] + ".o")
        flags = ["-DSPX_WG_DIAG", "-DSPX_WG_BURST=%d" % args.burst] if f == "conv_wgrad.hip" else []
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
                               *flags, "-c", os.path.join(csrc, f), "-o", o])
        objs.append(o)
    lib_path = os.path.join(out, "libspx_diag.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])

    import numpy as np
    import torch
    from spx import _lib
    _lib.LIB_PATH = lib_path
    from kbench import backbone8x_layers
    from pcdet_amd.datasets import synthetic
    from spx import ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    spec = synthetic.CONFIGS[2]
    geom = spec["geom"]
    batch = spec["batch"]
    b = synthetic.make_batch(2, batch)
    pts = torch.from_numpy(b["points"]).to(dev)
    vox = ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, geom["max_voxels"]["train"],
                       batch_size=batch, batch_col=0, xyz_col=1, feat_col=1, want_voxels=False)
    gs = synthetic.grid_size_of(geom)
    shape = [int(gs[2]) + 1, int(gs[1]), int(gs[0])]
    idx = vox["coords"]
    books = {}
    for name, cin, cout, ks, st, pd, ctype, key in backbone8x_layers(geom["num_point_features"]):
        if key not in books:
            books[key] = ops.subm_rulebook(idx, batch, shape, ks) if ctype == "subm" else \
                ops.conv_rulebook(idx, batch, shape, ks, st, pd)
        rb = books[key]
        if name == args.layer:
            break
        if ctype != "subm":
            idx, shape = rb.out_indices, rb.out_shape
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rb.n_in, cin, generator=g).to(dev)
    dout = torch.randn(rb.n_out, cout, generator=g).to(dev)
    wshape = (cout, *ks, cin)
    nblocks = 8 * 4096
    diag = torch.zeros(nblocks * 8, dtype=torch.int64, device=dev)
    lib.spx_diag_set.restype = ctypes.c_int
    lib.spx_diag_set.argtypes = [ctypes.c_void_p]
    for _ in range(20):                                  # warm clocks
        ops.conv_wgrad(x, dout, rb.pair, rb.ld, rb.n_out, wshape)
    torch.cuda.synchronize()
    assert lib.spx_diag_set(ctypes.c_void_p(diag.data_ptr())) == 0
    ops.conv_wgrad(x, dout, rb.pair, rb.ld, rb.n_out, wshape)
    torch.cuda.synchronize()
    d = diag.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 0] > 0]
    life, loop, steps, real, k, t0, t1, rows = [d[:, i].astype(np.float64) for i in range(8)]
    span = (t1.max() - t0.min())
    clk = np.median(life[real > 0] / real[real > 0]) * 100e6
    print("layer %s  blocks %d  kernel span %.0f cycles  in-kernel clock %.2f GHz -> %.1f us" % (
        args.layer, d.shape[0], span, clk / 1e9, span / clk * 1e6))
    print("block lifetime cycles: mean %.0f  median %.0f  max %.0f   (span/mean %.2f)" % (
        life.mean(), np.median(life), life.max(), span / life.mean()))
    print("steps per block: mean %.1f max %.0f;  cycles per step inside the step loops: mean %.0f  (MFMA-only floor %d)" % (
        steps.mean(), steps.max(), loop.sum() / max(steps.sum(), 1), 32 * ((cin + 15) // 16) * ((cout + 15) // 16)))
    print("share of block lifetime inside the step loops: %.2f" % (loop.sum() / life.sum()))
    start = t0 - t0.min()
    print("block start offsets (cycles): p50 %.0f  p90 %.0f  max %.0f" % tuple(np.percentile(start, [50, 90, 100])))
    for kk in sorted(set(k.astype(int))):
        m = k == kk
        if kk in (0, 4, 13, 22, 26):
            print("  k=%2d blocks %3d  rows/block %.0f steps/block %.0f  life %.0f  cyc/step %.0f" % (
                kk, m.sum(), rows[m].mean(), steps[m].mean(), life[m].mean(), loop[m].sum() / max(steps[m].sum(), 1)))


if __name__ == "__main__":
    main()
