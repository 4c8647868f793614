"""Dev tool (GPU box): per-wave timeline of k_conv_mfma on one layer of the cfg-2 batch.  Builds a DIAGNOSTIC copy of
libspx (csrc/conv_gemm.hip with -DSPX_CV_DIAG: cycle stamps per wave); the shipped library carries no stamps.

python tools/conv_diag.py [--layer conv3.1.0] [--env SPX_CONV_MT=1]"""
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
    ap.add_argument("--balanced", action="store_true", help="stamp the persistent balanced kernel (conv_balanced.hip)")
    ap.add_argument("--defs", default="", help="extra -D flags for conv_gemm.hip, comma separated (diag ablations)")
    args = ap.parse_args()
    csrc = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "csrc")
    out = "/tmp/spx_diag_cv"
    os.makedirs(out, exist_ok=True)
    objs = []
    for f in [f for f in os.listdir(csrc) if f.endswith(".hip")]:
        o = os.path.join(out, f[:-4] + ".o")
        flags = (["-DSPX_CV_DIAG"] + ["-D" + x for x in args.defs.split(",") if x]) if f in ("conv_gemm.hip", "conv_balanced.hip") else []
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
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rb.n_in, cin, generator=g).to(dev)
    w = (torch.randn(cout, *ks, cin, generator=g) / np.sqrt(rb.kvol * cin)).to(dev)
    wp = ops.pack_weight(w, 0)
    if args.balanced:
        return diag_balanced(lib, ops, x, wp, cout, cin, rb, dev)
    nwaves = (rb.n_out + 15) // 16 + 8
    diag = torch.zeros(nwaves * 4, dtype=torch.int64, device=dev)
    lib.spx_diag_set_conv.restype = ctypes.c_int
    lib.spx_diag_set_conv.argtypes = [ctypes.c_void_p]
    for _ in range(20):
        ops.conv_gemm(x, wp, cout, rb.kvol, rb.pair, rb.ld, rb.n_out)
    torch.cuda.synchronize()
    assert lib.spx_diag_set_conv(ctypes.c_void_p(diag.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_gemm(x, wp, cout, rb.kvol, rb.pair, rb.ld, rb.n_out)
    e1.record()
    torch.cuda.synchronize()
    d = diag.cpu().numpy().reshape(-1, 4).astype(np.float64)
    d = d[d[:, 0] > 0]
    raw = diag.cpu().numpy().reshape(-1, 4)
    raw = raw[raw[:, 0] > 0]
    life, t0, t1 = d[:, 0], d[:, 2], d[:, 3]
    units = (raw[:, 1] & 0xFF).astype(np.float64)
    drain = ((raw[:, 1] >> 8) & 0xFFFFF).astype(np.float64) * 64
    bar = ((raw[:, 1] >> 28) & 0xFFFFF).astype(np.float64) * 64
    mma = ((raw[:, 1] >> 48) & 0xFFFF).astype(np.float64) * 64
    span_us = (t1.max() - t0.min()) / 100.0
    clk = np.median(life / np.maximum(t1 - t0, 1)) * 100e6
    floor = 32 * (cin // 4) * (cout // 16)
    print("layer %s  waves %d  kernel span %.1f us (event %.1f us)  in-kernel clock %.2f GHz" % (
        args.layer, d.shape[0], span_us, e0.elapsed_time(e1) * 1e3, clk / 1e9))
    print("wave lifetime us: mean %.1f  p50 %.1f  p90 %.1f  max %.1f" % (
        (life / clk * 1e6).mean(), *np.percentile(life / clk * 1e6, [50, 90, 100])))
    print("units per wave: mean %.1f  max %.0f ; cycles per unit: mean %.0f (MFMA-only floor %d) ; total MFMA floor %.1f us" % (
        units.mean(), units.max(), life.sum() / units.sum(), floor, units.sum() * floor / 1024 / clk * 1e6))
    st = (t0 - t0.min()) / 100.0
    print("wave start offsets us: p10 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st, [10, 50, 75, 90, 100])))
    en = (t1 - t0.min()) / 100.0
    print("wave end   offsets us: p10 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f" % tuple(np.percentile(en, [10, 50, 75, 90, 100])))
    if drain.sum() > 0:
        late = (t0 - t0.min()) / 100.0 > 20.0
        for nm, m in (("first-round waves", ~late), ("late-start waves ", late)):
            if m.sum():
                print("%s %5d: life %.0f cyc = drain %.0f + barrier %.0f + mfma-phase %.0f + rest %.0f   (per wave; %d offsets)" % (
                    nm, m.sum(), life[m].mean(), drain[m].mean(), bar[m].mean(), mma[m].mean(),
                    (life[m] - drain[m] - bar[m] - mma[m]).mean(), rb.kvol))
    # resident waves over time
    ts = np.linspace(0, span_us, 11)
    print("resident waves at t:", " ".join("%.0f:%d" % (t, int(((st <= t) & (en > t)).sum())) for t in ts))


def diag_balanced(lib, ops, x, wp, cout, cin, rb, dev):
    import numpy as np
    import torch
    plan = ops.conv_plan(rb.pair, rb.ld, rb.kvol, rb.n_out)
    diag = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    lib.spx_diag_set_balanced.restype = ctypes.c_int
    lib.spx_diag_set_balanced.argtypes = [ctypes.c_void_p]
    for _ in range(20):
        ops.conv_gemm_balanced(x, wp, cout, rb.kvol, rb.pair, rb.ld, rb.n_out, plan)
    torch.cuda.synchronize()
    assert lib.spx_diag_set_balanced(ctypes.c_void_p(diag.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_gemm_balanced(x, wp, cout, rb.kvol, rb.pair, rb.ld, rb.n_out, plan)
    e1.record()
    torch.cuda.synchronize()
    hdr = plan[:4].cpu().numpy()
    d = diag.cpu().numpy().reshape(-1, 8).astype(np.float64)
    d = d[d[:, 0] > 0]
    life, units, act, bar, mma, t0, t1 = [d[:, i] for i in range(7)]
    clk = np.median(life / np.maximum(t1 - t0, 1)) * 100e6
    floor = 32 * (cin // 4) * (cout // 16)
    print("plan: workgroups %d  super-units %d  super-tiles %d ; waves stamped %d ; event %.1f us ; clock %.2f GHz" % (
        hdr[0], hdr[1], hdr[2], d.shape[0], e0.elapsed_time(e1) * 1e3, clk / 1e9))
    span = (t1.max() - t0.min()) / 100.0
    print("main kernel span %.1f us ; wave lifetime us mean %.1f p10 %.1f p90 %.1f max %.1f" % (
        span, (life / clk * 1e6).mean(), *np.percentile(life / clk * 1e6, [10, 90, 100])))
    print("per wave: super-units %.1f, active in %.1f (%.0f %%) ; cycles per super-unit %.0f = barrier %.0f + mfma phase %.0f + rest %.0f ; MFMA floor per active unit %d" % (
        units.mean(), act.mean(), 100 * act.sum() / units.sum(), life.sum() / units.sum(), bar.sum() / units.sum(),
        mma.sum() / units.sum(), (life - bar - mma).sum() / units.sum(), floor))
    print("MFMA floor of the launch %.1f us (active units x %d cycles / 1024 SIMDs)" % (act.sum() * floor / 1024 / clk * 1e6, floor))
    raw = diag.cpu().numpy().reshape(-1, 8)
    raw = raw[raw[:, 0] > 0]
    hw = raw[:, 7] & 0xFFFFFFFF
    xcc = (raw[:, 7] >> 32) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    import collections
    per_simd = collections.Counter(simd_key.tolist())
    per_cu = collections.Counter(cu_key.tolist())
    print("placement: %d CUs used, waves per CU %s ; %d SIMDs used, waves per SIMD %s" % (
        len(per_cu), dict(collections.Counter(per_cu.values())), len(per_simd), dict(collections.Counter(per_simd.values()))))
    lt = life / clk * 1e6
    print("   lifetime by XCC:", " ".join("%d:%.0f" % (x, lt[xcc == x].mean()) for x in sorted(set(xcc.tolist()))))
    print("   cycles per super-unit by XCC:", " ".join("%d:%.0f" % (x, life[xcc == x].sum() / units[xcc == x].sum()) for x in sorted(set(xcc.tolist()))))
    blk_id = np.arange(raw.shape[0]) // 4
    q4 = np.array_split(np.argsort(blk_id, kind="stable"), 8)
    print("   lifetime by workgroup-index octile:", " ".join("%.0f" % lt[i].mean() for i in q4))
    print("   active share by octile:", " ".join("%.2f" % (act[i].sum() / units[i].sum()) for i in q4))
    t_end = (t1 - t0.min()) / 100.0
    t_beg = (t0 - t0.min()) / 100.0
    simd_end = collections.defaultdict(float)
    simd_beg = collections.defaultdict(lambda: 1e9)
    simd_act = collections.defaultdict(float)
    for kk, e_, b_, a_ in zip(simd_key.tolist(), t_end.tolist(), t_beg.tolist(), act.tolist()):
        simd_end[kk] = max(simd_end[kk], e_)
        simd_beg[kk] = min(simd_beg[kk], b_)
        simd_act[kk] += a_
    ends = np.array(list(simd_end.values()))
    begs = np.array(list(simd_beg.values()))
    acts = np.array([simd_act[k_] for k_ in simd_end])
    print("   per SIMD: first wave starts at us p50 %.1f max %.1f ; last wave ends at us p10 %.1f p50 %.1f p90 %.1f max %.1f" % (
        np.percentile(begs, 50), begs.max(), *np.percentile(ends, [10, 50, 90, 100])))
    print("   per SIMD: active units min %.0f p50 %.0f max %.0f ; MFMA busy share until its own end: p10 %.2f p50 %.2f p90 %.2f" % (
        acts.min(), np.median(acts), acts.max(), *np.percentile(acts * floor / clk * 1e6 / ends, [10, 50, 90])))
    for nws in sorted(set(per_simd.values())):
        sel = np.array([per_simd[k] == nws for k in simd_key.tolist()])
        print("   waves on a SIMD holding %d waves: %5d  lifetime mean %.1f us" % (nws, sel.sum(), lt[sel].mean()))


if __name__ == "__main__":
    main()
