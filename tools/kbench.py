"""Per-layer microbenchmark of the sparse-conv kernels on the real rulebooks of a synthetic batch (dev tool).

python tools/kbench.py [--cfg 2] [--batch 4] [--iters 20] [--what fwd,dgrad,wgrad,rulebook]
Prints, per VoxelBackBone8x layer: rows, valid pairs, pairs/row, time, TFLOP/s (algorithmic 2*P*Cin*Cout), and the
compulsory-bytes GB/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pcdet_amd.datasets import synthetic  # noqa: E402
from spx import ops  # noqa: E402


def backbone8x_layers(input_channels=4):
    """(name, cin, cout, ksize, stride, padding, conv_type, indice_key) of the 12 sparse convolutions of VoxelBackBone8x, read
    off an instance of the module (dev tools use this table; the oracle keeps its own copy for the tests)."""
    import spx
    from pcdet_amd.config import AttrDict
    from pcdet_amd.models.backbones_3d import VoxelBackBone8x
    net = VoxelBackBone8x(AttrDict(), input_channels, [8, 8, 8])
    rows = []
    for name, m in net.named_modules():
        if isinstance(m, spx.conv.SparseConvolution):
            rows.append((name, m.in_channels, m.out_channels, tuple(m.kernel_size), tuple(m.stride), tuple(m.padding),
                         "subm" if m.subm else "spconv", m.indice_key))
    return rows


def grouped(pair, ld, K, n, window, flip=False):
    """perm (processing position -> table row), rows ordered by (window, offset mask), and the table in that order.
    window == -2: libspx's own spx_conv_group (one window), checked against the torch construction."""
    if window == -2:
        from spx import ops
        perm, pp = ops.conv_group(pair, ld, K, n)
        assert sorted(perm.tolist()) == list(range(n)), "spx_conv_group: perm is not a permutation"
        assert torch.equal(pp, pair[:K, :n][:, perm.long()]), "spx_conv_group: grouped table is not the table read through perm"
        t = timeit(lambda: ops.conv_group(pair, ld, K, n), 20)
        print("    spx_conv_group n=%d: %.1f us" % (n, t * 1e6))
        return perm, pp
    has = pair[:K, :n] >= 0
    w = (1 << torch.arange(K, device=pair.device, dtype=torch.int64))[:, None]
    mask = (has.to(torch.int64) * w).sum(0)
    win = torch.arange(n, device=pair.device, dtype=torch.int64) // (window if window > 0 else n)
    perm = torch.argsort((win << 27) | mask, stable=True).to(torch.int32)
    return perm, pair[:K, :n][:, perm.long()].contiguous()


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
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--what", default="fwd,dgrad,wgrad,rulebook")
    ap.add_argument("--layers", default="")
    ap.add_argument("--balanced", action="store_true", help="fwd/dgrad through the balanced persistent schedule")
    ap.add_argument("--ring", action="store_true", help="also time the round-3 ring schedule (csrc/conv_ring.hip), 64x64")
    ap.add_argument("--group", type=int, default=0,
                    help="with --balanced: also time the schedule over rows grouped by offset mask inside windows of this "
                         "many rows (-1: one window)")
    args = ap.parse_args()
    what = args.what.split(",")
    dev = torch.device("cuda:0")
    spec = synthetic.CONFIGS[args.cfg]
    geom = spec["geom"]
    batch = args.batch or spec["batch"]
    b = synthetic.make_batch(args.cfg, batch)
    pts = torch.from_numpy(b["points"]).to(dev)
    mv = spec.get("max_voxels", geom["max_voxels"]["train"])
    vox = ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, mv, batch_size=batch, batch_col=0,
                       xyz_col=1, feat_col=1, want_voxels=False)
    if "rulebook" in what:
        t = timeit(lambda: ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, mv, batch_size=batch,
                                        batch_col=0, xyz_col=1, feat_col=1, want_voxels=False), args.iters)
        print("voxelize+meanvfe: %d pts -> %d voxels  %.1f us  (%.1f Mpts/s)" % (pts.shape[0], vox["num_voxels"], t * 1e6,
                                                                               pts.shape[0] / t / 1e6))
    gs = synthetic.grid_size_of(geom)
    shape = [int(gs[2]) + 1, int(gs[1]), int(gs[0])]
    idx = vox["coords"]
    books = {}
    g = torch.Generator().manual_seed(0)
    tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0, rulebook=0.0, flops=0.0)
    for name, cin, cout, ks, st, pd, ctype, key in backbone8x_layers(geom["num_point_features"]):
        if key not in books:
            if ctype == "subm":
                books[key] = ops.subm_rulebook(idx, batch, shape, ks)
                fn = (lambda i=idx, s=list(shape), k=ks: ops.subm_rulebook(i, batch, s, k))
            else:
                books[key] = ops.conv_rulebook(idx, batch, shape, ks, st, pd)
                fn = (lambda i=idx, s=list(shape), k=ks, a=st, p=pd: ops.conv_rulebook(i, batch, s, k, a, p))
            if "rulebook" in what:
                t = timeit(fn, args.iters)
                tot["rulebook"] += t
                print("  rulebook %-12s n_in %7d -> n_out %7d  %8.1f us  (%.1f Mvoxels/s)" % (
                    key, idx.shape[0], books[key].n_out, t * 1e6, idx.shape[0] / t / 1e6))
        rb = books[key]
        if args.layers and name not in args.layers.split(","):
            if ctype != "subm":
                idx, shape = rb.out_indices, rb.out_shape
            continue
        K = rb.kvol
        P = int((rb.pair[:, :rb.n_out] >= 0).sum().item())
        x = torch.randn(rb.n_in, cin, generator=g).to(dev)
        w = (torch.randn(cout, *ks, cin, generator=g) / np.sqrt(K * cin)).to(dev)
        dout = torch.randn(rb.n_out, cout, generator=g).to(dev)
        wp, wt = ops.pack_weight(w, 0), ops.pack_weight(w, 1)
        flops = 2.0 * P * cin * cout
        nbytes = 4.0 * (rb.n_in * cin + rb.n_out * cout + K * cin * cout + K * rb.n_out)
        line = "%-12s %3d->%3d %-6s n_in %7d n_out %7d P %8d (%.1f/row)" % (name, cin, cout, ctype, rb.n_in, rb.n_out,
                                                                              P, P / max(rb.n_out, 1))
        bal_f = args.balanced and ops.balanced_ok(cin, cout, rb.n_out)
        bal_b = args.balanced and ops.balanced_ok(cout, cin, rb.n_in)
        if "fwd" in what:
            if bal_f:
                plan = ops.conv_plan(rb.pair, rb.ld, K, rb.n_out)
                t = timeit(lambda: ops.conv_gemm_balanced(x, wp, cout, K, rb.pair, rb.ld, rb.n_out, plan), args.iters)
                if args.group:
                    perm, pp = grouped(rb.pair, rb.ld, K, rb.n_out, args.group)
                    plang = ops.conv_plan(pp, rb.n_out, K, rb.n_out)
                    fg = (lambda: ops.conv_gemm_balanced(x, wp, cout, K, pp, rb.n_out, rb.n_out, plang, perm=perm))
                    err = float((fg() - ops.conv_gemm_balanced(x, wp, cout, K, rb.pair, rb.ld, rb.n_out, plan)).abs().max())
                    tg = timeit(fg, args.iters)
                    line += " | fwd grouped %7.1f us (units %d -> %d, err %.1e)" % (tg * 1e6, int(plan[1]), int(plang[1]), err)
            else:
                t = timeit(lambda: ops.conv_gemm(x, wp, cout, K, rb.pair, rb.ld, rb.n_out), args.iters)
            if args.ring and (cin, cout) in ((32, 32), (32, 64), (64, 32), (64, 64)) and K <= 31:
                ref = ops.conv_gemm(x, wp, cout, K, rb.pair, rb.ld, rb.n_out)
                rplan = ops.conv_ring_plan(rb.pair, rb.ld, K, rb.n_out)
                fr = (lambda: ops.conv_gemm_ring(x, wp, cout, K, rb.pair, rb.ld, rb.n_out, rplan))
                err = float((fr() - ref).abs().max())
                tr_ = timeit(fr, args.iters)
                tp = timeit(lambda: ops.conv_ring_plan(rb.pair, rb.ld, K, rb.n_out), args.iters)
                line += " | fwd ring %7.1f us %6.2f TF/s (err %.1e, plan %.1f us, timeouts %d)" % (
                    tr_ * 1e6, flops / tr_ / 1e12, err, tp * 1e6, int(rplan[2]))
                perm, pp = ops.conv_group(rb.pair, rb.ld, K, rb.n_out)
                gplan = ops.conv_ring_plan(pp, rb.n_out, K, rb.n_out)
                fg2 = (lambda: ops.conv_gemm_ring(x, wp, cout, K, pp, rb.n_out, rb.n_out, gplan, perm=perm))
                err = float((fg2() - ref).abs().max())
                tg2 = timeit(fg2, args.iters)
                out_s, st = ops.conv_gemm_ring(x, wp, cout, K, pp, rb.n_out, rb.n_out, gplan, perm=perm, want_stats=True)
                serr = float((st.double().sum(0)[0] - out_s.double().sum(0)).abs().max() / out_s.double().abs().sum(0).max())
                line += " | fwd ring grouped %7.1f us %6.2f TF/s (err %.1e, stats %.1e, units %d)" % (
                    tg2 * 1e6, flops / tg2 / 1e12, err, serr, int(gplan[1]))
            tot["fwd"] += t
            tot["flops"] += flops
            line += " | fwd %7.1f us %6.2f TF/s %6.0f GB/s" % (t * 1e6, flops / t / 1e12, nbytes / t / 1e9)
        if "dgrad" in what and cin >= 16:
            if bal_b:
                tb, ldb = (rb.pair, rb.ld) if rb.subm else (rb.pair_bwd, rb.pair_bwd.shape[1])
                planb = ops.conv_plan(tb, ldb, K, rb.n_in)
                f = (lambda: ops.conv_gemm_balanced(dout, wt, cin, K, tb, ldb, rb.n_in, planb, flip_k=rb.subm))
                if args.group:
                    permb, ppb = grouped(tb, ldb, K, rb.n_in, args.group)
                    plangb = ops.conv_plan(ppb, rb.n_in, K, rb.n_in)
                    fgb = (lambda: ops.conv_gemm_balanced(dout, wt, cin, K, ppb, rb.n_in, rb.n_in, plangb, flip_k=rb.subm,
                                                          perm=permb))
                    errb = float((fgb() - f()).abs().max())
                    line += " | dgrad grouped %7.1f us (err %.1e)" % (timeit(fgb, args.iters) * 1e6, errb)
            elif rb.subm:
                f = (lambda: ops.conv_gemm(dout, wt, cin, K, rb.pair, rb.ld, rb.n_in, flip_k=True))
            else:
                f = (lambda: ops.conv_gemm(dout, wt, cin, K, rb.pair_bwd, rb.pair_bwd.shape[1], rb.n_in))
            if args.ring and (cin, cout) in ((32, 32), (32, 64), (64, 32), (64, 64)) and K <= 31:
                tb, ldb = (rb.pair, rb.ld) if rb.subm else (rb.pair_bwd, rb.pair_bwd.shape[1])
                refb = ops.conv_gemm(dout, wt, cin, K, tb, ldb, rb.n_in, flip_k=rb.subm)
                rplanb = ops.conv_ring_plan(tb, ldb, K, rb.n_in)
                frb = (lambda: ops.conv_gemm_ring(dout, wt, cin, K, tb, ldb, rb.n_in, rplanb, flip_k=rb.subm))
                errb = float((frb() - refb).abs().max())
                line += " | dgrad ring %7.1f us (err %.1e)" % (timeit(frb, args.iters) * 1e6, errb)
            t = timeit(f, args.iters)
            tot["dgrad"] += t
            line += " | dgrad %7.1f us %6.2f TF/s" % (t * 1e6, flops / t / 1e12)
        if "wgrad" in what:
            t = timeit(lambda: ops.conv_wgrad(x, dout, rb.pair, rb.ld, rb.n_out, tuple(w.shape)), args.iters)
            tot["wgrad"] += t
            line += " | wgrad %7.1f us %6.2f TF/s" % (t * 1e6, flops / t / 1e12)
        print(line)
        if ctype != "subm":
            idx, shape = rb.out_indices, rb.out_shape
    print("TOTAL per batch of %d: fwd %.3f ms (%.2f TF/s)  dgrad %.3f ms  wgrad %.3f ms  rulebooks %.3f ms" % (
        batch, tot["fwd"] * 1e3, tot["flops"] / max(tot["fwd"], 1e-9) / 1e12, tot["dgrad"] * 1e3, tot["wgrad"] * 1e3,
        tot["rulebook"] * 1e3))


if __name__ == "__main__":
    main()
