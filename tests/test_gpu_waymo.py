"""BASELINE configs[2] (Waymo, 2 x 80k voxels, fwd+bwd) and configs[4] (Waymo 5-frame concat, 300k voxels, forward) on the
GPU: the Waymo model (tools/cfgs/waymo_models/second.yaml: C = 5 point features, 0.1 m voxels, 1504 x 1504 x 40 grid,
150k voxel cap — reference tools/cfgs/dataset_configs/waymo_dataset.yaml:6,73-79), the kernels and dispatch branches that
only these sizes reach (k_conv_mfma MT=2 at >= 2^18 rows, the plan scan beyond its LDS prefix, bitmaps of 47 M cells).

 * detector-level parity against the oracle backend at a size the CPU affords (synthetic cfg 6: 2 x 8k voxels on the full
   Waymo grid): indices exact, features 1e-4, boxes 1e-3, gradients as in tests/test_gpu_model.py;
 * full sizes: bit-exact voxeliser and rulebooks against the C oracle, one 16->16 / 32->32 / 64->64 layer forward, dgrad and
   wgrad against the oracle, and size-independent properties (rulebook symmetry / inverse maps, linearity, bitwise re-run,
   finite outputs, stage shapes)."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-5


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    assert err <= tol * scale, "max|err| %.3e > %.1e * %.3g" % (err, tol, scale)


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _build_waymo(cfg_id, seed=0, training=True):
    from pcdet_amd.config import AttrDict, cfg_from_yaml_file
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models import build_network
    cfg = cfg_from_yaml_file(os.path.join(ROOT, "tsm-det-pointcloud-_amd/tools/cfgs/waymo_models/second.yaml"), AttrDict())
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training, cfg_id=cfg_id)
    cfg.MODEL.VFE.VOXELIZE.MAX_NUMBER_OF_VOXELS = ds.max_voxels
    torch.manual_seed(seed)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    g = torch.Generator().manual_seed(seed + 1)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return cfg, ds, model


def _batch(ds, n=2):
    b = ds.collate_batch([ds[i] for i in range(n)])
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}


def _to(bd, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in bd.items()}


# ------------------------------------------------------------------------------------------ Waymo model vs oracle backend

def test_waymo_model_is_the_c5_waymo_yaml():
    _cfg, ds, model = _build_waymo(6)
    assert ds.point_feature_encoder.num_point_features == 5 and list(ds.grid_size) == [1504, 1504, 40]
    assert model.backbone_3d.conv_input[0].in_channels == 5
    assert model.backbone_3d.sparse_shape == [41, 1504, 1504]
    assert ds.max_voxels["train"] == 150000


def test_waymo_detector_eval_forward_parity():
    from oracle.cpu_backend import use_oracle_backend
    _cfg, ds, model = _build_waymo(6)
    model.eval()
    ref = copy.deepcopy(model)
    bd_c = _batch(ds)
    with torch.no_grad(), use_oracle_backend():
        for m in ref.module_list:
            bd_c = m(bd_c)
    dev = _dev()
    model.to(dev)
    bd_g = _to(_batch(ds), dev)
    with torch.no_grad():
        for m in model.module_list:
            bd_g = m(bd_g)
    assert bd_g["voxel_features"].shape[1] == 5
    assert torch.equal(bd_g["voxel_coords"].cpu(), bd_c["voxel_coords"])
    assert _rel(bd_g["voxel_features"], bd_c["voxel_features"]) < 1e-6
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        tg, tc = bd_g["multi_scale_3d_features"][k], bd_c["multi_scale_3d_features"][k]
        assert torch.equal(tg.indices.cpu(), tc.indices) and tg.spatial_shape == tc.spatial_shape
        assert _rel(tg.features, tc.features) < 1e-4, k
    eg, ec = bd_g["encoded_spconv_tensor"], bd_c["encoded_spconv_tensor"]
    assert torch.equal(eg.indices.cpu(), ec.indices) and eg.spatial_shape == [2, 188, 188]
    assert _rel(eg.features, ec.features) < 1e-4
    assert _rel(bd_g["spatial_features"], bd_c["spatial_features"]) < 1e-4
    assert _rel(bd_g["batch_cls_preds"], bd_c["batch_cls_preds"]) < 1e-4
    assert float((bd_g["batch_box_preds"].cpu() - bd_c["batch_box_preds"]).abs().max()) < 1e-3     # boxes: 1e-3


def test_waymo_detector_train_step_parity():
    """One training step (train-mode BatchNorm) of the Waymo model, GPU vs oracle backend: loss and loss terms 1e-4, BN
    running statistics 1e-4, gradients within the summation-order bar of tests/test_gpu_model.py (2e-2 global)."""
    from oracle.cpu_backend import use_oracle_backend
    _cfg, ds, model = _build_waymo(6, seed=3)
    model.train()
    ref = copy.deepcopy(model)
    with use_oracle_backend():
        ret_c, tb_c, _ = ref(_batch(ds))
        ret_c["loss"].backward()
    dev = _dev()
    model.to(dev)
    ret_g, tb_g, _ = model(_to(_batch(ds), dev))
    ret_g["loss"].backward()
    lg, lc = float(ret_g["loss"].detach()), float(ret_c["loss"].detach())
    assert abs(lg - lc) < 1e-4 * abs(lc)
    for k in tb_c:
        assert abs(float(tb_g[k]) - float(tb_c[k])) < 1e-4 * max(1.0, abs(float(tb_c[k]))), k
    pg, pc = dict(model.named_parameters()), dict(ref.named_parameters())
    num = den = 0.0
    worst = ("", 0.0)
    for name, p in pc.items():
        assert pg[name].grad is not None, name
        d2 = float((pg[name].grad.cpu().double() - p.grad.double()).pow(2).sum())
        n2 = float(p.grad.double().pow(2).sum())
        r = (d2 / max(n2, 1e-30)) ** 0.5
        worst = (name, r) if r > worst[1] else worst
        num, den = num + d2, den + n2
    print("waymo train-BN: worst per-parameter rel-L2 %s %.2e ; global rel-L2 %.2e" % (worst[0], worst[1], (num / den) ** 0.5))
    assert worst[1] < 5e-2, worst
    assert (num / den) ** 0.5 < 2e-2
    bg, bc = dict(model.named_buffers()), dict(ref.named_buffers())
    for name in bc:
        if name.endswith("running_mean") or name.endswith("running_var"):
            assert _rel(bg[name], bc[name]) < 1e-4, name


# ------------------------------------------------------------------------------------------ full sizes

def _frames(cfg_id, n):
    from pcdet_amd.datasets import synthetic as syn
    return [syn.make_frame(cfg_id, i)["points"] for i in range(n)]


def _voxelize_both(orc, frames, geom, max_voxels):
    """GPU batched voxelise vs per-frame C oracle: bit-exact coords / counts / voxels; returns (gpu dict, idx, shape)."""
    from pcdet_amd.datasets import synthetic as syn
    from spx import ops
    dev = _dev()
    pts = np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1) for b, f in enumerate(frames)], 0)
    out = ops.voxelize(torch.from_numpy(pts).to(dev), geom["point_cloud_range"], geom["voxel_size"], 5, max_voxels,
                       batch_size=len(frames), batch_col=0, xyz_col=1, feat_col=1)
    vs, cs, ns = [], [], []
    for b, f in enumerate(frames):
        v, c, n = orc.voxelize(f, geom["point_cloud_range"], geom["voxel_size"], 5, max_voxels)
        vs.append(v), cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1)), ns.append(n)
    v, c, n = np.concatenate(vs), np.concatenate(cs), np.concatenate(ns)
    assert out["num_voxels"] == v.shape[0]
    assert np.array_equal(out["coords"].cpu().numpy(), c)
    assert np.array_equal(out["num_points"].cpu().numpy(), n)
    assert np.array_equal(out["voxels"].cpu().numpy(), v)
    shape = [int(x) for x in (syn.grid_size_of(geom)[::-1] + [1, 0, 0])]
    return out, c, shape


GEOMS = [("subm", (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("sp", (3, 3, 3), (2, 2, 2), (1, 1, 1)),
         ("subm", (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("sp", (3, 3, 3), (2, 2, 2), (1, 1, 1)),
         ("subm", (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("sp", (3, 3, 3), (2, 2, 2), (0, 1, 1)),
         ("subm", (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("sp", (3, 1, 1), (2, 1, 1), (0, 0, 0))]


def _rulebook_chain(orc, idx_np, shape, batch, exact):
    """The 8 rule tables of VoxelBackBone8x chained through the four levels.  exact: compare every table with the C oracle
    bit for bit; always: the size-independent properties.  Returns the GPU rulebooks."""
    from spx import ops
    dev = _dev()
    books = []
    for kind, k, s, p in GEOMS:
        d_idx = torch.from_numpy(idx_np).to(dev)
        K = k[0] * k[1] * k[2]
        if kind == "subm":
            rb = ops.subm_rulebook(d_idx, batch, shape, k, want_cnt=True)
            pair = rb.pair[:, :rb.n_out]
            assert torch.equal(pair[K // 2], torch.arange(rb.n_out, device=dev, dtype=torch.int32))
            for kk in range(K):                                   # symmetry: pair[k][o] = i <=> pair[K-1-k][i] = o
                o = (pair[kk] >= 0).nonzero()[:, 0]
                assert torch.equal(pair[K - 1 - kk][pair[kk][o].long()], o.int())
            assert int((pair >= 0).sum()) == int(rb.cnt.sum())
            if exact:
                pair_o, cnt_o = orc.subm_rulebook(idx_np, shape, k)
                assert np.array_equal(pair.cpu().numpy(), pair_o) and np.array_equal(rb.cnt.cpu().numpy(), cnt_o)
        else:
            rb = ops.conv_rulebook(d_idx, batch, shape, k, s, p, want_cnt=True)
            oi = rb.out_indices.long()
            key = ((oi[:, 0] * rb.out_shape[0] + oi[:, 1]) * rb.out_shape[1] + oi[:, 2]) * rb.out_shape[2] + oi[:, 3]
            assert bool((key[1:] > key[:-1]).all())               # canonical ascending order, no duplicates
            pf, pb = rb.pair[:, :rb.n_out], rb.pair_bwd
            assert bool(((pf >= 0).sum(0) >= 1).all())
            assert int((pf >= 0).sum()) == int((pb >= 0).sum()) == int(rb.cnt.sum())
            for kk in range(K):                                   # forward and backward tables are inverse maps
                i = (pb[kk] >= 0).nonzero()[:, 0]
                assert torch.equal(pf[kk][pb[kk][i].long()], i.int())
            if exact:
                oi_o, pf_o, pb_o, cnt_o, oshape = orc.conv_rulebook(idx_np, shape, k, s, p)
                assert rb.n_out == oi_o.shape[0] and rb.out_shape == oshape
                assert np.array_equal(rb.out_indices.cpu().numpy(), oi_o)
                assert np.array_equal(pf.cpu().numpy(), pf_o) and np.array_equal(pb.cpu().numpy(), pb_o)
                assert np.array_equal(rb.cnt.cpu().numpy(), cnt_o)
            idx_np, shape = rb.out_indices.cpu().numpy(), rb.out_shape
        books.append(rb)
    return books


def _layer_vs_oracle(orc, rb, cin, cout, seed, wgrad=True):
    """One sparse conv layer on rule table rb through the product dispatch (spx.functional: balanced / grouped schedule
    where it applies): forward, dgrad, wgrad against the oracle; bitwise identical on a second run."""
    from spx import functional as F_, ops
    dev = _dev()
    K = rb.kvol
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rb.n_in, cin, generator=g)
    w = torch.randn(cout, K, cin, generator=g) / np.sqrt(K * cin)
    dout = torch.randn(rb.n_out, cout, generator=g)
    pair_np = rb.pair[:, :rb.n_out].cpu().numpy()
    xg = x.to(dev).requires_grad_(True)
    wg = w.view(cout, *rb.ksize, cin).to(dev).requires_grad_(True)
    out = F_.sparse_conv(xg, wg, None, rb)
    out.backward(dout.to(dev))
    _close(out.detach().cpu().numpy(), orc.conv_fwd_gemm(x.numpy(), w.view(cout, *rb.ksize, cin).numpy(), pair_np))
    _close(xg.grad.cpu().numpy(), orc.conv_dgrad(dout.numpy(), w.view(cout, *rb.ksize, cin).numpy(), pair_np, rb.n_in))
    if wgrad:
        _close(wg.grad.cpu().numpy(), orc.conv_wgrad(x.numpy(), dout.numpy(), pair_np, tuple(wg.shape)), tol=1e-4)
    x2 = x.to(dev).requires_grad_(True)
    w2 = w.view(cout, *rb.ksize, cin).to(dev).requires_grad_(True)
    out2 = F_.sparse_conv(x2, w2, None, rb)
    out2.backward(dout.to(dev))
    assert torch.equal(out2, out) and torch.equal(x2.grad, xg.grad) and torch.equal(w2.grad, wg.grad)
    # the plain one-tile-per-wave kernel on the same table agrees (the dispatch above may have taken another one)
    ref = ops.conv_gemm(x.to(dev), ops.pack_weight(wg.detach(), 0), cout, K, rb.pair, rb.ld, rb.n_out)
    assert _rel(out, ref) < 2e-6


def test_cfg3_waymo_full_size_voxelise_rulebooks_layers(orc):
    """BASELINE configs[2]: 2 Waymo frames x 80k voxels.  Voxeliser and all 8 rule tables bit-exact against the C oracle;
    a 16->16, a 32->32 and a 64->64 layer forward / dgrad / wgrad against the oracle on their real tables."""
    from pcdet_amd.datasets import synthetic as syn
    _out, idx_np, shape = _voxelize_both(orc, _frames(3, 2), syn.WAYMO, 150000)
    assert idx_np.shape[0] == 160000
    books = _rulebook_chain(orc, idx_np, shape, 2, exact=True)
    _layer_vs_oracle(orc, books[0], 16, 16, 1)       # subm1
    _layer_vs_oracle(orc, books[2], 32, 32, 2)       # subm2 (~ 200k rows)
    _layer_vs_oracle(orc, books[3], 32, 64, 3)       # spconv3
    _layer_vs_oracle(orc, books[4], 64, 64, 4)       # subm3
    _layer_vs_oracle(orc, books[7], 64, 128, 5)      # spconv_down2, kernel (3,1,1)


def test_cfg3_waymo_full_size_train_step_properties():
    """The Waymo detector at 2 x 80k voxels, one fwd + bwd step: stage shapes, finite loss / gradients / boxes, and the
    whole step bitwise reproducible (every libspx reduction has a fixed order; MIOpen's dense tail is deterministic for
    a fixed solver choice)."""
    _cfg, ds, model = _build_waymo(3, seed=5)
    dev = _dev()
    model.to(dev).train()
    bd = _to(_batch(ds), dev)

    def step():
        model.zero_grad(set_to_none=True)
        b = dict(bd)
        ret, tb, _ = model(b)
        ret["loss"].backward()
        gs = torch.cat([p.grad.flatten() for p in model.backbone_3d.parameters()])
        return float(ret["loss"]), gs, b

    l1, g1, b1 = step()
    ms = b1["multi_scale_3d_features"]
    assert ms["x_conv1"].features.shape == (160000, 16) and ms["x_conv1"].spatial_shape == [41, 1504, 1504]
    assert ms["x_conv2"].spatial_shape == [21, 752, 752] and ms["x_conv2"].features.shape[1] == 32
    assert ms["x_conv3"].spatial_shape == [11, 376, 376] and ms["x_conv4"].spatial_shape == [5, 188, 188]
    enc = b1["encoded_spconv_tensor"]
    assert enc.spatial_shape == [2, 188, 188] and enc.features.shape[1] == 128
    assert b1["spatial_features"].shape == (2, 256, 188, 188)
    assert np.isfinite(l1) and bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    sparse_before = {k: v.features.detach().clone() for k, v in ms.items()}
    l2, g2, b2 = step()
    for k, v in b2["multi_scale_3d_features"].items():         # sparse stack forward: bitwise (BN running stats do not feed it)
        assert torch.equal(v.features, sparse_before[k]), k
    assert _rel(g2, g1) < 1e-5 and abs(l2 - l1) < 1e-6 * abs(l1)
    model.eval()
    with torch.no_grad():
        pred, _rec = model(dict(bd))
    assert len(pred) == 2 and all(bool(torch.isfinite(p["pred_boxes"]).all()) for p in pred)


def test_cfg5_waymo_concat_300k_voxels(orc):
    """BASELINE configs[4]: one 5-frame Waymo concatenation, ~600k points -> 300k voxels (cap 400k).  Voxeliser bit-exact;
    the 8 rule tables: subm1 + spconv2 bit-exact against the oracle, all of them through the size-independent properties;
    16->16 and 64->64 layers (>= 2^18 rows: the MT=2 dispatch branch of the 64-channel kernels) forward / dgrad / wgrad vs the oracle + bitwise
    re-run; forward of the Waymo detector finite with the expected stage shapes."""
    from pcdet_amd.datasets import synthetic as syn
    from spx import ops
    out, idx_np, shape = _voxelize_both(orc, _frames(5, 1), syn.WAYMO, 400000)
    assert idx_np.shape[0] == 300000
    books = _rulebook_chain(orc, idx_np, shape, 1, exact=False)
    dev = _dev()
    pair_o, cnt_o = orc.subm_rulebook(idx_np, shape, (3, 3, 3))
    assert np.array_equal(books[0].pair[:, :books[0].n_out].cpu().numpy(), pair_o)
    oi_o, pf_o, pb_o, cnt2_o, _osh = orc.conv_rulebook(idx_np, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    assert np.array_equal(books[1].out_indices.cpu().numpy(), oi_o)
    assert np.array_equal(books[1].pair[:, :books[1].n_out].cpu().numpy(), pf_o)
    assert np.array_equal(books[1].pair_bwd.cpu().numpy(), pb_o)
    assert books[0].n_out >= (1 << 18)
    _layer_vs_oracle(orc, books[0], 16, 16, 11)
    _layer_vs_oracle(orc, books[0], 64, 64, 12)      # 64 -> 64 on the 300k-row level-1 table: MT=2 / balanced + grouped
    # densify of a 300k-row tensor is exact
    f = torch.randn(books[7].n_out, 128, generator=torch.Generator().manual_seed(1))
    dense = ops.densify(f.to(dev), books[7].out_indices, 1, books[7].out_shape)
    assert np.array_equal(dense.cpu().numpy(), orc.densify(f.numpy(), books[7].out_indices.cpu().numpy(), 1, books[7].out_shape))


def test_cfg5_waymo_detector_forward_finite():
    _cfg, ds, model = _build_waymo(5, training=False)
    dev = _dev()
    model.to(dev).eval()
    bd = _to(_batch(ds, 1), dev)
    with torch.no_grad():
        for m in model.module_list:
            bd = m(bd)
    assert bd["voxel_coords"].shape[0] == 300000
    ms = bd["multi_scale_3d_features"]
    assert ms["x_conv1"].features.shape == (300000, 16) and ms["x_conv4"].spatial_shape == [5, 188, 188]
    for k, v in ms.items():
        assert bool(torch.isfinite(v.features).all()), k
    assert bd["batch_box_preds"].shape[0] == 1 and bool(torch.isfinite(bd["batch_box_preds"]).all())
    assert bool(torch.isfinite(bd["batch_cls_preds"]).all())


@pytest.mark.parametrize("n,cases", [(600_000, ((16, 16, False),)), (1_100_000, ((16, 16, False), (64, 64, True)))])
def test_million_row_dispatch_branches(n, cases, orc):
    """Row counts beyond anything the synthetic configs produce: 2^19 <= rows < 2^20 switches the low-channel layers of
    spx_conv_gemm from the latency form to two 16-row tiles per wave (MT = 2), >= 2^20 to four (MT = 4), and a plan over
    more than 786k rows leaves the LDS-resident prefix of k_plan_scan.  Synthetic rule table (random neighbours, 10 %
    density), 16->16 through spx_conv_gemm and 64->64 through the balanced schedule, against the numpy oracle; bitwise
    identical on a second launch."""
    from spx import ops
    dev = _dev()
    K = 27
    rs = np.random.RandomState(9)
    pair_np = np.where(rs.rand(K, n) < 0.10, rs.randint(0, n, size=(K, n)), -1).astype(np.int32)
    pair_np[13] = np.arange(n, dtype=np.int32)                    # the centre offset of a submanifold table
    pair = torch.from_numpy(pair_np).to(dev)
    g = torch.Generator().manual_seed(4)
    for cs, cd, balanced in cases:
        x = torch.randn(n, cs, generator=g)
        w = torch.randn(cd, K, cs, generator=g) / np.sqrt(K * cs * 0.1)
        wp = ops.pack_weight(w.to(dev), 0)
        xg = x.to(dev)
        if balanced:
            plan = ops.conv_plan(pair, n, K, n)
            assert int(plan[2]) == (n + 63) // 64 and int(plan[2]) + 1 > 12 * 1024      # beyond the LDS prefix of the plan scan
            out = ops.conv_gemm_balanced(xg, wp, cd, K, pair, n, n, plan)
            again = ops.conv_gemm_balanced(xg, wp, cd, K, pair, n, n, plan)
        else:
            out = ops.conv_gemm(xg, wp, cd, K, pair, n, n)
            again = ops.conv_gemm(xg, wp, cd, K, pair, n, n)
        assert torch.equal(out, again)
        ref = orc.conv_fwd_gemm(x.numpy(), w.view(cd, 3, 3, 3, cs).numpy(), pair_np)
        _close(out.cpu().numpy(), ref)
        del out, again, xg, ref
