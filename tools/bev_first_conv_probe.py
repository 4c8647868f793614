"""bev_first_conv_probe.py — dev tool: the first convolution of BaseBEVBackbone (reference base_bev_backbone.py:27-34:
ZeroPad2d(1) + Conv2d(256, 128, 3)) reads the densified sparse tensor of HeightCompression (height_compression.py:21-23),
i.e. mostly zeros.  As a sparse conv over the encoded tensor (kernel (D,3,3), padding (0,1,1): z folds into channels
exactly as the BEV view does) it has ~8x fewer FLOPs.  This script times the sparse form (rulebook, forward, dgrad, wgrad,
densify) next to the dense MIOpen convolution on the same data and checks that the results agree."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsm-det-pointcloud-_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from kbench import timeit  # noqa: E402
from pcdet_amd.datasets import synthetic  # noqa: E402
from spx import ops  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    spec = synthetic.CONFIGS[cfg]
    geom, batch = spec["geom"], spec["batch"]
    pts = []
    for b in range(batch):
        f = synthetic.make_frame(cfg, b)
        p = torch.from_numpy(f["points"]).to(dev)
        pts.append(torch.cat([torch.full((p.shape[0], 1), float(b), device=dev), p], 1))
    pts = torch.cat(pts, 0)
    vox = ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, 400000, batch_size=batch, batch_col=0,
                       xyz_col=1, feat_col=1)
    idx = vox["coords"]
    shape = [int(x) for x in (np.asarray(synthetic.grid_size_of(geom))[::-1] + [1, 0, 0])]
    for k, s, p in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
                    ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
        rb = ops.conv_rulebook(idx, batch, shape, k, s, p)
        idx, shape = rb.out_indices, rb.out_shape
    D, H, W = shape
    n, C, CO = idx.shape[0], 128, 128
    print("encoded tensor: %d rows, shape %s, batch %d -> BEV [%d, %d, %d, %d], pixel occupancy %.3f" % (
        n, shape, batch, batch, C * D, H, W, len(torch.unique((idx[:, 0].long() * H + idx[:, 2]) * W + idx[:, 3])) / (batch * H * W)))
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(n, C, generator=g).to(dev)
    w2 = (torch.randn(CO, C * D, 3, 3, generator=g) / np.sqrt(9 * C * D)).to(dev)
    # dense reference
    dense = ops.densify(feats, idx, batch, shape, channels_last=True)                 # logical [B, C, D, H, W]
    bev = dense.permute(0, 1, 2, 3, 4).reshape(batch, C * D, H, W) if False else dense.reshape(batch, C * D, H, W)
    bev = bev.contiguous(memory_format=torch.channels_last)
    w2cl = w2.contiguous(memory_format=torch.channels_last)
    yd = torch.nn.functional.conv2d(bev, w2cl, None, 1, 1)
    t_dense = timeit(lambda: torch.nn.functional.conv2d(bev, w2cl, None, 1, 1), 20)
    gy = torch.randn(yd.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    bev_r = bev.clone().requires_grad_(True)
    w_r = w2cl.clone().requires_grad_(True)

    def dense_bwd():
        y = torch.nn.functional.conv2d(bev_r, w_r, None, 1, 1)
        bev_r.grad = w_r.grad = None
        y.backward(gy)
    t_dense_fb = timeit(dense_bwd, 10)
    flops_d = 2.0 * 9 * C * D * CO * batch * H * W
    print("dense  conv2d fwd %7.1f us (%.1f TF/s)   fwd+bwd %7.1f us" % (t_dense * 1e6, flops_d / t_dense / 1e12, t_dense_fb * 1e6))
    # sparse form
    ks, st, pd = (D, 3, 3), (1, 1, 1), (0, 1, 1)
    rb = ops.conv_rulebook(idx, batch, shape, ks, st, pd)
    dout = torch.randn(rb.n_out, CO, generator=g).to(dev)
    t_rb = timeit(lambda: ops.conv_rulebook(idx, batch, shape, ks, st, pd), 10)
    K = rb.kvol
    P = int((rb.pair[:, :rb.n_out] >= 0).sum())
    w3 = w2.view(CO, C, D, 3, 3).permute(0, 2, 3, 4, 1).contiguous()                  # [co, d, ky, kx, c]
    wp, wt = ops.pack_weight(w3, 0), ops.pack_weight(w3, 1)
    ys = ops.conv_gemm(feats, wp, CO, K, rb.pair, rb.ld, rb.n_out)
    yd_rows = yd.permute(0, 2, 3, 1)[rb.out_indices[:, 0].long(), rb.out_indices[:, 2].long(), rb.out_indices[:, 3].long()]
    err = float((ys - yd_rows).abs().max() / yd_rows.abs().max())
    covered = torch.zeros(batch, H, W, dtype=torch.bool, device=dev)
    covered[rb.out_indices[:, 0].long(), rb.out_indices[:, 2].long(), rb.out_indices[:, 3].long()] = True
    rest = float(yd.permute(0, 2, 3, 1)[~covered].abs().max()) if (~covered).any() else 0.0
    print("sparse form: n_out %d (%.3f of the pixels), K %d, P %d, rel err vs dense %.2e, |dense| outside the output set %.1e" % (
        rb.n_out, rb.n_out / (batch * H * W), K, P, err, rest))
    flops = 2.0 * P * C * CO
    t_f = timeit(lambda: ops.conv_gemm(feats, wp, CO, K, rb.pair, rb.ld, rb.n_out), 20)
    t_d = timeit(lambda: ops.conv_gemm(dout, wt, C, K, rb.pair_bwd, rb.pair_bwd.shape[1], rb.n_in), 20)
    t_w = timeit(lambda: ops.conv_wgrad(feats, dout, rb.pair, rb.ld, rb.n_out, tuple(w3.shape)), 20)
    out_shape = rb.out_shape
    t_dn = timeit(lambda: ops.densify(ys, rb.out_indices, batch, out_shape, channels_last=True), 20)
    # balanced two-half kernel (plan over the table), and the same over rows grouped by offset mask
    os.environ.setdefault("SPX_CONV_BALANCED_SHAPES", "128x128")
    plan = ops.conv_plan(rb.pair, rb.ld, K, rb.n_out)
    yb = ops.conv_gemm_balanced(feats, wp, CO, K, rb.pair, rb.ld, rb.n_out, plan)
    t_fb = timeit(lambda: ops.conv_gemm_balanced(feats, wp, CO, K, rb.pair, rb.ld, rb.n_out, plan), 20)
    perm, grouped = ops.conv_group(rb.pair, rb.ld, K, rb.n_out)
    plang = ops.conv_plan(grouped, rb.n_out, K, rb.n_out)
    yg = ops.conv_gemm_balanced(feats, wp, CO, K, grouped, rb.n_out, rb.n_out, plang, perm=perm)
    t_fg = timeit(lambda: ops.conv_gemm_balanced(feats, wp, CO, K, grouped, rb.n_out, rb.n_out, plang, perm=perm), 20)
    t_grp = timeit(lambda: ops.conv_group(rb.pair, rb.ld, K, rb.n_out), 20)
    t_pl = timeit(lambda: ops.conv_plan(rb.pair, rb.ld, K, rb.n_out), 20)
    nb_ = rb.pair_bwd.shape[1]
    planb = ops.conv_plan(rb.pair_bwd, nb_, K, rb.n_in)
    db_ref = ops.conv_gemm(dout, wt, C, K, rb.pair_bwd, nb_, rb.n_in)
    db_bal = ops.conv_gemm_balanced(dout, wt, C, K, rb.pair_bwd, nb_, rb.n_in, planb)
    t_db = timeit(lambda: ops.conv_gemm_balanced(dout, wt, C, K, rb.pair_bwd, nb_, rb.n_in, planb), 20)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print("balanced 128x128: fwd %6.1f us (err %.1e, units %d) | fwd grouped %6.1f us (err %.1e, units %d; group %5.1f us, plan %5.1f us)"
          " | dgrad %6.1f us (err %.1e)" % (t_fb * 1e6, rel(yb, ys), int(plan[1]), t_fg * 1e6, rel(yg, ys), int(plang[1]),
                                            t_grp * 1e6, t_pl * 1e6, t_db * 1e6, rel(db_bal, db_ref)))
    print("sparse rulebook %6.1f us | fwd %6.1f us (%.1f TF/s) | dgrad %6.1f us | wgrad %6.1f us | densify out %6.1f us | sum %.1f us" % (
        t_rb * 1e6, t_f * 1e6, flops / t_f / 1e12, t_d * 1e6, t_w * 1e6, t_dn * 1e6, (t_rb + t_f + t_d + t_w + 2 * t_dn) * 1e6))


if __name__ == "__main__":
    main()
