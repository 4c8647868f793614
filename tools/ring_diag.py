"""Dev tool (GPU box): per-wave timeline of k_conv_ring (csrc/conv_ring.hip) on one layer of the cfg-2 batch.  Builds a
DIAGNOSTIC copy of libspx (conv_ring.hip with -DSPX_RING_DIAG: cycle stamps per consumer wave; the other objects are the
shipped ones from csrc/build); the shipped library carries no stamps.

python tools/ring_diag.py [--layer conv3.1.0] [--defs A,B] [--plain]"""
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
    ap.add_argument("--defs", default="", help="extra -D flags for conv_ring.hip, comma separated (ablations)")
    ap.add_argument("--plain", action="store_true", help="rows in table order (default: grouped by offset mask)")
    ap.add_argument("--time-only", action="store_true", help="print the launch time only (ablation builds)")
    args = ap.parse_args()
    csrc = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "csrc")
    out = "/tmp/spx_diag_ring"
    os.makedirs(out, exist_ok=True)
    o = os.path.join(out, "conv_ring.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
                           *([] if args.time_only else ["-DSPX_RING_DIAG"]), *["-D" + x for x in args.defs.split(",") if x], "-c",
                           os.path.join(csrc, "conv_ring.hip"), "-o", o])
    objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build"))
            if f.endswith(".o") and f != "conv_ring.o"]
    lib_path = os.path.join(out, "libspx_diag.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, o, *objs])

    import numpy as np
    import torch
    from spx import _lib
    _lib.LIB_PATH = lib_path
    from kbench import backbone8x_layers, timeit
    from pcdet_amd.datasets import synthetic
    from spx import ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    spec = synthetic.CONFIGS[2]
    geom, batch = spec["geom"], spec["batch"]
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
    K = rb.kvol
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rb.n_in, cin, generator=g).to(dev)
    w = (torch.randn(cout, *ks, cin, generator=g) / np.sqrt(K * cin)).to(dev)
    wp = ops.pack_weight(w, 0)
    if args.plain:
        pair, ld, perm = rb.pair, rb.ld, None
    else:
        perm, pair = ops.conv_group(rb.pair, rb.ld, K, rb.n_out)
        ld = rb.n_out
    plan = ops.conv_ring_plan(pair, ld, K, rb.n_out)
    f = (lambda: ops.conv_gemm_ring(x, wp, cout, K, pair, ld, rb.n_out, plan, perm=perm))
    t = timeit(f, 20)
    if args.time_only:
        print("layer %s  defs [%s]  event %.1f us" % (args.layer, args.defs, t * 1e6))
        return
    ncw = 12
    diag = torch.zeros((256 * ncw * 16,), dtype=torch.int64, device=dev)
    lib.spx_diag_set_ring.restype = ctypes.c_int
    lib.spx_diag_set_ring.argtypes = [ctypes.c_void_p]
    assert lib.spx_diag_set_ring(ctypes.c_void_p(diag.data_ptr())) == 0
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    d = diag.cpu().numpy().reshape(256, ncw, 16).astype(np.float64)
    hdr = plan[:32].cpu().numpy()
    life, units, proto, mma, issue = d[..., 0], d[..., 1], d[..., 2], d[..., 3], d[..., 4]
    t0, t1 = d[..., 5], d[..., 6]
    span = (t1.max() - t0.min()) / 100.0
    clock = life.sum() / ((t1 - t0).sum() / 100.0) / 1e3           # cycles per us / 1e3 = GHz
    print("layer %s  rows %d  tiles %d  units %d  rounds/range %s  event %.1f us  span %.1f us  clock %.2f GHz  timeouts %d" % (
        args.layer, rb.n_out, hdr[0], hdr[1], list(hdr[8:16]), t * 1e6, span, clock, hdr[2]))
    print("MFMA floor %.1f us at that clock (units x 2048 cycles / 1024 SIMDs)" % (hdr[1] * 2048.0 / 1024 / clock / 1e3))
    us = lambda c: c / clock / 1e3
    print("wave lifetime us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f ; start spread %.1f us" % (
        us(life.mean()), us(np.percentile(life, 10)), us(np.percentile(life, 50)), us(np.percentile(life, 90)), us(life.max()),
        (t0.max() - t0.min()) / 100.0))
    passed, tail, first, lastu = d[..., 10], d[..., 11], d[..., 12], d[..., 13]
    print("per wave: units mean %.1f max %.0f ; per-unit cycles: issue %.0f  signal-behind %.0f  wait-own-slot %.0f  lds+mfma %.0f  (sum %.0f ; floor 2048)" % (
        units.mean(), units.max(), issue.sum() / units.sum(), passed.sum() / units.sum(), proto.sum() / units.sum(),
        mma.sum() / units.sum(), (issue.sum() + passed.sum() + proto.sum() + mma.sum()) / units.sum()))
    tot = life.sum()
    print("share of wave lifetime: start->first step %.2f  issue %.2f  signal-behind %.2f  wait-own-slot %.2f  lds+mfma %.2f  epilogue %.2f  tail %.2f  other %.2f" % (
        first.sum() / tot, issue.sum() / tot, passed.sum() / tot, proto.sum() / tot, mma.sum() / tot, d[..., 9].sum() / tot,
        tail.sum() / tot, 1 - (first.sum() + issue.sum() + passed.sum() + proto.sum() + mma.sum() + d[..., 9].sum() + tail.sum()) / tot))
    print("start -> first step us: mean %.1f max %.1f ; start -> last MFMA us: mean %.1f p90 %.1f max %.1f ; tail us mean %.1f" % (
        us(first.mean()), us(first.max()), us(lastu.mean()), us(np.percentile(lastu, 90)), us(lastu.max()), us(tail.mean())))
    heavy = units >= np.percentile(units, 90)
    print("heaviest 10%% of the waves: units %.1f ; per-unit cycles: issue %.0f signal-behind %.0f wait-own-slot %.0f lds+mfma %.0f ; lifetime %.1f us" % (
        units[heavy].mean(), issue[heavy].sum() / units[heavy].sum(), passed[heavy].sum() / units[heavy].sum(),
        proto[heavy].sum() / units[heavy].sum(), mma[heavy].sum() / units[heavy].sum(), us(life[heavy].mean())))
    print("end of round 0 (us after the wave's start): mean %.1f p10 %.1f p90 %.1f max %.1f" % (
        us(d[..., 8].mean()), us(np.percentile(d[..., 8], 10)), us(np.percentile(d[..., 8], 90)), us(d[..., 8].max())))
    # per SIMD: waves w, w+4, w+8 are dealt to one SIMD bin by the plan; check the placement and the balance
    hw = d[..., 7].astype(np.int64) & 0xFFFFFFFF
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    same = sum(int(len(set(simd[wg, s::4])) == 1) for wg in range(256) for s in range(4))
    print("SIMD groups (waves w, w+4, w+8) that really share a SIMD: %d of 1024" % same)
    ub = np.stack([units[:, s::4].sum(1) for s in range(4)], 1)       # units per planned SIMD bin
    print("units per planned SIMD bin: mean %.1f min %.0f max %.0f ; per workgroup: mean %.1f min %.0f max %.0f" % (
        ub.mean(), ub.min(), ub.max(), units.sum(1).mean(), units.sum(1).min(), units.sum(1).max()))
    wl = life.max(1)
    print("workgroup lifetime us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % (us(np.percentile(wl, 10)), us(np.percentile(wl, 50)),
                                                                            us(np.percentile(wl, 90)), us(wl.max())))
    byx = [us(life[x::8].max(1).mean()) for x in range(8)]
    print("workgroup lifetime by blockIdx & 7: " + " ".join("%.0f" % v for v in byx))


if __name__ == "__main__":
    main()
