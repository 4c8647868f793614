"""HIP-vs-oracle parity for every kernel family, called through the C ABI (spx.ops -> libspx.so).

Bars (BASELINE.json north_star): voxel indices / rulebooks BIT-EXACT; fp32 features within a stated
tolerance: |err| <= 2e-5 * max(1, max|ref|) for a single conv (fp32 MFMA == k-ordered fmaf chain; the
oracle sums in another order, so round-off differs by a few ulp per accumulated term).
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 2e-5


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close(a, b, tol=TOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    assert err <= tol * scale, "max|err| %.3e > %.1e * %.3g" % (err, tol, scale)


def _rulebook(ops, d_idx, batch, shape, k, s, p, subm):
    if subm:
        return ops.subm_rulebook(d_idx, batch, shape, k, want_cnt=True)
    return ops.conv_rulebook(d_idx, batch, shape, k, s, p, want_cnt=True)


# ------------------------------------------------------------------------------------------ golden fixtures

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "dense_conv_*.npz"))))
def test_golden_dense_conv(path, orc):
    """Committed dense-F.conv3d vectors: index set exact, values / dgrad / wgrad to fp32 round-off."""
    from spx import ops
    d = np.load(path)
    dev = _dev()
    idx = torch.from_numpy(d["idx"]).to(dev)
    batch, shape = int(d["batch"]), [int(x) for x in d["shape"]]
    k, s, p = [int(x) for x in d["k"]], [int(x) for x in d["s"]], [int(x) for x in d["p"]]
    rb = _rulebook(ops, idx, batch, shape, k, s, p, bool(d["subm"]))
    assert rb.n_out == d["out_idx"].shape[0]
    assert np.array_equal(rb.out_indices.cpu().numpy(), d["out_idx"])
    assert rb.out_shape == [int(x) for x in d["out_spatial"]]
    w = torch.from_numpy(d["w"]).to(dev)
    feat = torch.from_numpy(d["feat"]).to(dev)
    cout, cin, K = w.shape[0], w.shape[-1], rb.kvol
    out = ops.conv_gemm(feat, ops.pack_weight(w, 0), cout, K, rb.pair, rb.ld, rb.n_out)
    _close(out.cpu().numpy(), d["out"])
    dout = torch.from_numpy(d["dout"]).to(dev)
    wt = ops.pack_weight(w, 1)
    if rb.subm:
        din = ops.conv_gemm(dout, wt, cin, K, rb.pair, rb.ld, rb.n_in, flip_k=True)
    else:
        din = ops.conv_gemm(dout, wt, cin, K, rb.pair_bwd, rb.pair_bwd.shape[1], rb.n_in)
    _close(din.cpu().numpy(), d["dfeat"])
    dw = ops.conv_wgrad(feat, dout, rb.pair, rb.ld, rb.n_out, tuple(w.shape))
    _close(dw.cpu().numpy(), d["dw"], tol=5e-5)


# ------------------------------------------------------------------------------------------ dynamic voxelisation

@pytest.mark.parametrize("cfg_id,nframes", [(0, 3), (2, 4)])
def test_dynamic_voxelize_matches_oracle(cfg_id, nframes, orc):
    """Row a5' (DynamicMeanVFE): unique cells in key order, coords (b, z, y, x), point -> voxel map: exact; per-voxel means
    bit-exact as well (both sum in point order in fp32).  Includes out-of-range points and a point ON the upper bound."""
    from pcdet_amd.datasets import synthetic as syn
    from spx import ops
    geom = syn.CONFIGS[cfg_id]["geom"]
    rng, vs = geom["point_cloud_range"], geom["voxel_size"]
    b = syn.make_batch(cfg_id, nframes)
    pts = b["points"].copy()
    extra = np.array([[0, rng[0] - 1.0, 0.0, 0.0, 0.5], [1, rng[3], 0.0, 0.0, 0.5], [0, 1.0, rng[4] + 5.0, 0.0, 0.1],
                      [nframes - 1, 10.0, 1.0, rng[5], 0.2], [0, 10.0, 1.0, -1.0, 0.3], [0, 10.02, 1.01, -1.01, 0.7]], np.float32)
    pts = np.concatenate([pts, extra[:, :pts.shape[1]]], 0)
    rs = np.random.RandomState(3)
    pts = pts[rs.permutation(pts.shape[0])]                     # frames interleaved: nothing relies on contiguity
    f_o, c_o, inv_o = orc.dynamic_voxelize(pts, rng, vs, batch_size=nframes)
    out = ops.dynamic_voxelize(torch.from_numpy(pts).to(_dev()), rng, vs, batch_size=nframes)
    assert out["num_voxels"] == c_o.shape[0]
    assert np.array_equal(out["coords"].cpu().numpy(), c_o)
    assert np.array_equal(out["inverse"].cpu().numpy(), inv_o)
    assert np.array_equal(out["features"].cpu().numpy(), f_o)
    # keys ascending = the order torch.unique gives the reference
    g = [int(round((rng[3 + j] - rng[j]) / vs[j])) for j in range(3)]
    c = out["coords"].cpu().numpy().astype(np.int64)
    key = ((c[:, 0] * g[0] + c[:, 3]) * g[1] + c[:, 2]) * g[2] + c[:, 1]
    assert (np.diff(key) > 0).all()
    # capped call: the count still reports every voxel, rows beyond the cap are dropped, mapped points say so
    cap = c_o.shape[0] // 2
    o2 = ops.dynamic_voxelize(torch.from_numpy(pts).to(_dev()), rng, vs, batch_size=nframes, max_voxels=cap)
    assert int(o2["d_num_voxels"].item()) == c_o.shape[0] and o2["coords"].shape[0] == cap
    assert np.array_equal(o2["features"].cpu().numpy(), f_o[:cap])
    inv2 = o2["inverse"].cpu().numpy()
    assert np.array_equal(inv2[inv_o < cap], inv_o[inv_o < cap]) and (inv2[inv_o >= cap] == -1).all()


def test_dynamic_mean_vfe_module_and_empty_input():
    from pcdet_amd.config import AttrDict
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models.backbones_3d.vfe import DynamicMeanVFE
    from spx import ops
    ds = SyntheticDataset(cfg_id=0)
    vfe = DynamicMeanVFE(AttrDict(), 4, ds.voxel_size, ds.grid_size, ds.point_cloud_range)
    b = ds.collate_batch([ds[0], ds[1]])
    bd = vfe({"points": torch.from_numpy(b["points"]).to(_dev()), "batch_size": 2})
    assert bd["voxel_features"].shape[1] == 4 and bd["voxel_coords"].shape == (bd["voxel_features"].shape[0], 4)
    assert vfe.get_output_feature_dim() == 4 and bd["voxel_coords"].dtype == torch.int32
    far = torch.full((7, 5), 1e6, device=_dev())
    far[:, 0] = 0
    out = ops.dynamic_voxelize(far, ds.point_cloud_range, ds.voxel_size, batch_size=1)
    assert out["num_voxels"] == 0 and (out["inverse"] == -1).all()


# ------------------------------------------------------------------------------------------ rulebooks

def _frame_indices(orc, cfg_id, nframes):
    from pcdet_amd.datasets import synthetic as syn
    geom = syn.CONFIGS[cfg_id]["geom"]
    idx = []
    for b in range(nframes):
        f = syn.make_frame(cfg_id, b)
        _v, c, _n = orc.voxelize(f["points"], geom["point_cloud_range"], geom["voxel_size"], 5, 400000)
        idx.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    shape = [int(x) for x in (syn.grid_size_of(geom)[::-1] + [1, 0, 0])]
    return np.concatenate(idx, 0), shape


BACKBONE_GEOMS = [
    ("subm", (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("sp", (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ("sp", (3, 3, 3), (2, 2, 2), (0, 1, 1)),
    ("sp", (3, 1, 1), (2, 1, 1), (0, 0, 0)),
]


@pytest.mark.parametrize("cfg_id,nframes", [(1, 2), (2, 2)])
def test_rulebooks_bit_exact_vs_oracle(cfg_id, nframes, orc):
    """Every rulebook geometry of VoxelBackBone8x (spconv_backbone.py:86-122), chained through the four
    levels, bit-exact against the oracle: out indices (canonical order), pair tables, per-offset counts."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, cfg_id, nframes)
    dev = _dev()
    for kind, k, s, p in BACKBONE_GEOMS:
        d_idx = torch.from_numpy(idx_np).to(dev)
        if kind == "subm":
            rb = ops.subm_rulebook(d_idx, nframes, shape, k, want_cnt=True)
            pair_o, cnt_o = orc.subm_rulebook(idx_np, shape, k)
            assert np.array_equal(rb.pair[:, :rb.n_out].cpu().numpy(), pair_o)
            assert np.array_equal(rb.cnt.cpu().numpy(), cnt_o)
        else:
            rb = ops.conv_rulebook(d_idx, nframes, shape, k, s, p, want_cnt=True)
            oi, pf, pb, cnt_o, oshape = orc.conv_rulebook(idx_np, shape, k, s, p)
            assert rb.n_out == oi.shape[0] and rb.out_shape == oshape
            assert np.array_equal(rb.out_indices.cpu().numpy(), oi)
            assert np.array_equal(rb.pair[:, :rb.n_out].cpu().numpy(), pf)
            assert np.array_equal(rb.pair_bwd.cpu().numpy(), pb)
            assert np.array_equal(rb.cnt.cpu().numpy(), cnt_o)
            idx_np, shape = oi, oshape  # feed the next level


@pytest.mark.parametrize("cfg_id,nframes", [(1, 2), (2, 4), (5, 1)])
def test_subm_rulebook_symmetric_probe_equals_full_probe(cfg_id, nframes, orc):
    """SPX_ROWS_UNIQUE (what the modules pass: spconv's one-row-per-cell contract): half of the table probed, every hit written
    twice — bit-identical to the full probe (which the test above pins to the oracle), for the reference's 3x3x3 tables in
    voxeliser order and canonical order, other odd kernels, a dilation, a device-side live count, rows outside the grid, and
    a 1x1x1 / an even kernel (which take the full probe)."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, cfg_id, nframes)
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    canon = torch.from_numpy(idx_np).to(dev)
    shuffled = canon[torch.randperm(canon.shape[0], generator=g).to(dev)]
    for d_idx in (canon, shuffled):
        for k, dil in (((3, 3, 3), (1, 1, 1)), ((3, 1, 1), (1, 1, 1)), ((1, 3, 5), (1, 1, 1)), ((3, 3, 3), (1, 2, 2)),
                       ((1, 1, 1), (1, 1, 1)), ((2, 2, 2), (1, 1, 1))):
            full = ops.subm_rulebook(d_idx, nframes, shape, k, dil)
            sym = ops.subm_rulebook(d_idx, nframes, shape, k, dil, unique=True)
            assert torch.equal(sym.pair, full.pair), (k, dil)
    n = canon.shape[0]
    d_n = torch.tensor([n - 777], dtype=torch.int64, device=dev)
    full = ops.subm_rulebook(shuffled, nframes, shape, (3, 3, 3), d_n=d_n)
    sym = ops.subm_rulebook(shuffled, nframes, shape, (3, 3, 3), d_n=d_n, unique=True)
    assert torch.equal(sym.pair[:, :n - 777], full.pair[:, :n - 777])
    outside = shuffled.clone()
    outside[5::97, 1] = shape[0] + 3                      # rows outside the grid: in no cell, neighbours of nobody
    outside[11::89, 3] = -2
    full = ops.subm_rulebook(outside, nframes, shape, (3, 3, 3))
    sym = ops.subm_rulebook(outside, nframes, shape, (3, 3, 3), unique=True)
    assert torch.equal(sym.pair, full.pair)
    ops.check_status(dev)


def test_rulebook_edge_cases(orc):
    from spx import ops
    dev = _dev()
    shape = [5, 6, 7]
    # empty input
    e = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    rb = ops.subm_rulebook(e, 1, shape, (3, 3, 3))
    assert rb.n_out == 0
    rb = ops.conv_rulebook(e, 1, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    assert rb.n_out == 0
    # single voxel in a corner, and a fully dense tiny grid (every offset valid in the interior)
    for idx_np in (np.array([[0, 0, 0, 0]], np.int32),
                   np.array([[1, 4, 5, 6]], np.int32),
                   np.stack(np.meshgrid([0, 1], range(5), range(6), range(7), indexing="ij"), -1).reshape(-1, 4)
                   .astype(np.int32)):
        d_idx = torch.from_numpy(idx_np).to(dev)
        rb = ops.subm_rulebook(d_idx, 2, shape, (3, 3, 3), want_cnt=True)
        pair_o, cnt_o = orc.subm_rulebook(idx_np, shape)
        assert np.array_equal(rb.pair[:, :rb.n_out].cpu().numpy(), pair_o)
        for k, s, p in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (1, 1, 1), (1, 1, 1)),
                        ((2, 2, 2), (2, 2, 2), (0, 0, 0)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
            rb = ops.conv_rulebook(d_idx, 2, shape, k, s, p, want_cnt=True)
            oi, pf, pb, cnt_o, _ = orc.conv_rulebook(idx_np, shape, k, s, p)
            assert np.array_equal(rb.out_indices.cpu().numpy(), oi)
            assert np.array_equal(rb.pair[:, :rb.n_out].cpu().numpy(), pf)
            assert np.array_equal(rb.pair_bwd.cpu().numpy(), pb)
            assert np.array_equal(rb.cnt.cpu().numpy(), cnt_o)


def test_rulebook_properties_full_size(orc):
    """BASELINE cfg 3 size (Waymo, 2 x 80k voxels): size-independent properties, no oracle needed.
    subm symmetry pair[k][o]=i <=> pair[K-1-k][i]=o; strided fwd/bwd tables are inverse maps; out keys
    strictly ascending; every output has >= 1 pair; pair counts agree between tables."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 3, 2)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    rb = ops.subm_rulebook(d_idx, 2, shape, (3, 3, 3), want_cnt=True)
    pair = rb.pair[:, :rb.n_out]
    K = rb.kvol
    rows = torch.arange(rb.n_out, device=dev, dtype=torch.int32)
    assert torch.equal(pair[K // 2], rows)
    for k in range(K):
        o = (pair[k] >= 0).nonzero()[:, 0]
        i = pair[k][o].long()
        assert torch.equal(pair[K - 1 - k][i], o.int())
    rb2 = ops.conv_rulebook(d_idx, 2, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), want_cnt=True)
    oi = rb2.out_indices.long()
    key = ((oi[:, 0] * rb2.out_shape[0] + oi[:, 1]) * rb2.out_shape[1] + oi[:, 2]) * rb2.out_shape[2] + oi[:, 3]
    assert bool((key[1:] > key[:-1]).all())
    pf, pb = rb2.pair[:, :rb2.n_out], rb2.pair_bwd
    assert bool(((pf >= 0).sum(0) >= 1).all())
    assert int((pf >= 0).sum()) == int((pb >= 0).sum()) == int(rb2.cnt.sum())
    for k in range(K):
        i = (pb[k] >= 0).nonzero()[:, 0]
        assert torch.equal(pf[k][pb[k][i].long()], i.int())


# ------------------------------------------------------------------------------------------ arithmetic

CHANNELS = [(4, 16), (5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (7, 9),
            (48, 24)]


@pytest.mark.parametrize("cin,cout", CHANNELS)
def test_conv_fwd_dgrad_wgrad_vs_oracle(cin, cout, orc):
    """Every channel pair of VoxelBackBone8x (+ MFMA-untiled ones) on KITTI-crop geometry, subm and strided."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 1, 1)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    feat = torch.randn(idx_np.shape[0], cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) * (1.0 / np.sqrt(27 * cin))
    for subm in (True, False):
        rb = (ops.subm_rulebook(d_idx, 1, shape, (3, 3, 3)) if subm else
              ops.conv_rulebook(d_idx, 1, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1)))
        pair_np = rb.pair[:, :rb.n_out].cpu().numpy()
        out = ops.conv_gemm(feat.to(dev), ops.pack_weight(w.to(dev), 0), cout, 27, rb.pair, rb.ld, rb.n_out)
        ref = orc.conv_fwd(feat.numpy(), w.numpy(), pair_np, acc64=True)
        _close(out.cpu().numpy(), ref)
        # fused epilogue: scale/shift/relu
        sc = torch.rand(cout, generator=g) + 0.5
        sh = torch.randn(cout, generator=g)
        out2 = ops.conv_gemm(feat.to(dev), ops.pack_weight(w.to(dev), 0), cout, 27, rb.pair, rb.ld, rb.n_out,
                             scale=sc.to(dev), shift=sh.to(dev), relu=True)
        _close(out2.cpu().numpy(), np.maximum(ref * sc.numpy() + sh.numpy(), 0))
        dout = torch.randn(rb.n_out, cout, generator=g)
        wt = ops.pack_weight(w.to(dev), 1)
        if subm:
            din = ops.conv_gemm(dout.to(dev), wt, cin, 27, rb.pair, rb.ld, rb.n_in, flip_k=True)
        else:
            din = ops.conv_gemm(dout.to(dev), wt, cin, 27, rb.pair_bwd, rb.pair_bwd.shape[1], rb.n_in)
        _close(din.cpu().numpy(), orc.conv_dgrad(dout.numpy(), w.numpy(), pair_np, rb.n_in))
        ref_dw = orc.conv_wgrad(feat.numpy(), dout.numpy(), pair_np, tuple(w.shape))
        dw = ops.conv_wgrad(feat.to(dev), dout.to(dev), rb.pair, rb.ld, rb.n_out, tuple(w.shape))
        _close(dw.cpu().numpy(), ref_dw, tol=1e-4)
        # fixed summation order: a second launch is bitwise identical
        assert torch.equal(dw, ops.conv_wgrad(feat.to(dev), dout.to(dev), rb.pair, rb.ld, rb.n_out, tuple(w.shape)))


def test_conv_full_size_linearity_and_oracle(orc):
    """cfg 2 size (4 x 16k voxels, 64->64 subm3-like): linearity conv(a*x+y) = a*conv(x)+conv(y) and a
    direct oracle comparison (numpy gather-GEMM-scatter)."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 2, 4)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    rb = ops.subm_rulebook(d_idx, 4, shape, (3, 3, 3))
    g = torch.Generator().manual_seed(7)
    x = torch.randn(rb.n_in, 64, generator=g).to(dev)
    y = torch.randn(rb.n_in, 64, generator=g).to(dev)
    w = (torch.randn(64, 3, 3, 3, 64, generator=g) / np.sqrt(27 * 64)).to(dev)
    wp = ops.pack_weight(w, 0)
    fx = ops.conv_gemm(x, wp, 64, 27, rb.pair, rb.ld, rb.n_out)
    fy = ops.conv_gemm(y, wp, 64, 27, rb.pair, rb.ld, rb.n_out)
    fz = ops.conv_gemm(2.5 * x + y, wp, 64, 27, rb.pair, rb.ld, rb.n_out)
    _close(fz.cpu().numpy(), (2.5 * fx + fy).cpu().numpy(), tol=5e-5)
    ref = orc.conv_fwd_gemm(x.cpu().numpy(), w.cpu().numpy(), rb.pair[:, :rb.n_out].cpu().numpy())
    _close(fx.cpu().numpy(), ref)
    # determinism: bitwise identical on a second launch
    assert torch.equal(fx, ops.conv_gemm(x, wp, 64, 27, rb.pair, rb.ld, rb.n_out))
    # weight gradient at full size: oracle comparison, bilinearity in dout, fixed summation order
    dout = torch.randn(rb.n_out, 64, generator=g).to(dev)
    dw = ops.conv_wgrad(x, dout, rb.pair, rb.ld, rb.n_out, tuple(w.shape))
    ref_dw = orc.conv_wgrad(x.cpu().numpy(), dout.cpu().numpy(), rb.pair[:, :rb.n_out].cpu().numpy(), tuple(w.shape))
    _close(dw.cpu().numpy(), ref_dw, tol=1e-4)
    dw2 = ops.conv_wgrad(x, 2.0 * dout, rb.pair, rb.ld, rb.n_out, tuple(w.shape))
    assert torch.equal(dw2, 2.0 * dw)                                                   # scaling by 2 is exact in fp32
    assert torch.equal(dw, ops.conv_wgrad(x, dout, rb.pair, rb.ld, rb.n_out, tuple(w.shape)))


def test_conv_balanced_schedule_matches_tile_per_wave(orc):
    """spx_conv_gemm_balanced (persistent grid, equal MFMA work per wave, cut tiles summed head + tail by the fix-up
    kernel) against spx_conv_gemm on the same tables: forward, flipped (dgrad of a submanifold conv), strided tables in
    both directions, with and without the fused epilogue; bitwise reproducible from launch to launch."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 2, 4)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    g = torch.Generator().manual_seed(11)
    sub = ops.subm_rulebook(d_idx, 4, shape, (3, 3, 3))
    strd = ops.conv_rulebook(d_idx, 4, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    cases = [(sub.pair, sub.ld, sub.n_out, sub.n_in, False), (sub.pair, sub.ld, sub.n_in, sub.n_out, True),
             (strd.pair, strd.ld, strd.n_out, strd.n_in, False),
             (strd.pair_bwd, strd.pair_bwd.shape[1], strd.n_in, strd.n_out, False)]
    for (cs, cd) in ((64, 64), (32, 64), (64, 32), (32, 32), (128, 128)):     # 128x128: the two-half kernel
        w = (torch.randn(cd, 3, 3, 3, cs, generator=g) / np.sqrt(27 * cs)).to(dev)
        wp = ops.pack_weight(w, 0)
        for pair, ld, n_dst, n_src, flip in cases:
            x = torch.randn(n_src, cs, generator=g).to(dev)
            plan = ops.conv_plan(pair, ld, 27, n_dst)
            hdr = plan[:4].cpu().numpy()
            assert 1 <= hdr[0] <= 1024 and hdr[1] >= 8 * hdr[0] and hdr[2] == (n_dst + 63) // 64
            ref = ops.conv_gemm(x, wp, cd, 27, pair, ld, n_dst, flip_k=flip)
            out = ops.conv_gemm_balanced(x, wp, cd, 27, pair, ld, n_dst, plan, flip_k=flip)
            assert _rel_t(out, ref) < 2e-6
            assert torch.equal(out, ops.conv_gemm_balanced(x, wp, cd, 27, pair, ld, n_dst, plan, flip_k=flip))
        sc = (torch.rand(cd, generator=g) + 0.5).to(dev)
        sh = torch.randn(cd, generator=g).to(dev)
        pair, ld, n_dst, n_src, flip = cases[0]
        x = torch.randn(n_src, cs, generator=g).to(dev)
        plan = ops.conv_plan(pair, ld, 27, n_dst)
        ref = ops.conv_gemm(x, wp, cd, 27, pair, ld, n_dst, scale=sc, shift=sh, relu=True)
        out = ops.conv_gemm_balanced(x, wp, cd, 27, pair, ld, n_dst, plan, scale=sc, shift=sh, relu=True)
        assert _rel_t(out, ref) < 2e-6 and float(out.min()) >= 0.0


def _group_sort_key(mask, K, n, live):
    """The order spx_conv_group promises (csrc/conv_group.hip): window of the row, then the 11-bit group key of its offset
    mask (3x3x3: the nine in-plane offsets bit by bit + any-below + any-above); dead rows keep their place at the end."""
    dev = mask.device
    rows = torch.arange(n, device=dev)
    if K <= 11:
        g = mask
    elif K % 3 == 0 and K // 3 <= 9:
        P = K // 3
        low, mid, up = mask & ((1 << P) - 1), (mask >> P) & ((1 << P) - 1), mask >> (2 * P)
        g = mid | ((low != 0).long() << 9) | ((up != 0).long() << 10)
    else:
        g = mask >> (K - 11)
    W = max(8, (live + 4095) // 4096)
    W = (W + 7) // 8 * 8
    wsz = (live + W - 1) // W
    key = ((rows // max(wsz, 1)) << 11) | g
    return torch.where(rows < live, key, (1 << 40) + rows)


@pytest.mark.parametrize("frames,n_live", [(4, None), (1, None), (2, 9000)])
def test_conv_grouped_row_order(frames, n_live, orc):
    """spx_conv_group: perm is the stable sort of the rows by (window, 11-bit group key of the offset mask) — windows of at
    most 4 096 live rows, at least eight — (a permutation; dead rows last, in place), the grouped
    table is the table read through perm, the plan over it holds fewer (super-tile, offset) units, and the balanced
    conv over the grouped rows returns the rows of the ungrouped result (same per-row sums) — forward, flipped, with
    the fused epilogue, and with a device-side live-row count."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 2, frames)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    g = torch.Generator().manual_seed(23)
    d_n = None if n_live is None else torch.tensor([n_live], dtype=torch.int64, device=dev)
    sub = ops.subm_rulebook(d_idx, frames, shape, (3, 3, 3), d_n=d_n)
    n, K = sub.n_out, 27
    live = n if n_live is None else n_live
    perm, grouped = ops.conv_group(sub.pair, sub.ld, K, n, d_n)
    pair = sub.pair[:, :n]
    mask = ((pair >= 0).to(torch.int64) << torch.arange(K, device=dev)[:, None]).sum(0)
    key = _group_sort_key(mask, K, n, live)
    assert torch.equal(perm.long(), torch.argsort(key, stable=True))
    want = pair[:, perm.long()].clone()
    want[:, live:] = -1
    assert torch.equal(grouped, want)
    plan_c = ops.conv_plan(sub.pair, sub.ld, K, n, d_n)
    plan_g = ops.conv_plan(grouped, n, K, n, d_n)
    assert int(plan_g[1]) < int(plan_c[1])                              # fewer MFMA units to issue
    for cs, cd in ((64, 64), (32, 32), (128, 128)):
        w = (torch.randn(cd, 3, 3, 3, cs, generator=g) / np.sqrt(27 * cs)).to(dev)
        wp = ops.pack_weight(w, 0)
        x = torch.randn(n, cs, generator=g).to(dev)
        sc, sh = (torch.rand(cd, generator=g) + 0.5).to(dev), torch.randn(cd, generator=g).to(dev)
        for flip, kw in ((False, {}), (True, {}), (False, dict(scale=sc, shift=sh, relu=True))):
            ref = ops.conv_gemm(x, wp, cd, K, sub.pair, sub.ld, n, flip_k=flip, d_n_dst=d_n, **kw)
            out = ops.conv_gemm_balanced(x, wp, cd, K, grouped, n, n, plan_g, flip_k=flip, d_n_dst=d_n, perm=perm, **kw)
            assert _rel_t(out[:live], ref[:live]) < 2e-6
            again = ops.conv_gemm_balanced(x, wp, cd, K, grouped, n, n, plan_g, flip_k=flip, d_n_dst=d_n, perm=perm, **kw)
            assert torch.equal(out[:live], again[:live])


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 200, 1000, 5000])
def test_conv_balanced_small_and_ragged_sizes(n):
    """The balanced schedule at sizes where most workgroups have nothing to do, rows do not fill the last 16-row tile /
    64-row super-tile, and (n = 1) a single super-unit exists: same results as the one-tile-per-wave kernel."""
    from spx import ops
    dev = _dev()
    g = torch.Generator().manual_seed(100 + n)
    # n distinct voxels in a small grid -> a submanifold rulebook with a few neighbours per row
    side = max(4, int(np.ceil((2.5 * n) ** (1.0 / 3.0))))
    lin = torch.randperm(side ** 3, generator=g)[:n].sort()[0]
    idx = torch.stack([torch.zeros_like(lin), lin // (side * side), (lin // side) % side, lin % side], 1).int().to(dev)
    rb = ops.subm_rulebook(idx, 1, [side, side, side], (3, 3, 3))
    w = (torch.randn(64, 3, 3, 3, 64, generator=g) / 40.0).to(dev)
    wp = ops.pack_weight(w, 0)
    x = torch.randn(n, 64, generator=g).to(dev)
    plan = ops.conv_plan(rb.pair, rb.ld, 27, n)
    for flip in (False, True):
        ref = ops.conv_gemm(x, wp, 64, 27, rb.pair, rb.ld, n, flip_k=flip)
        out = ops.conv_gemm_balanced(x, wp, 64, 27, rb.pair, rb.ld, n, plan, flip_k=flip)
        assert _rel_t(out, ref) < 2e-6
    # the two-half 128 -> 128 kernel (two virtual workgroups per physical one) on the same ragged tables
    x2 = torch.randn(n, 128, generator=g).to(dev)
    w2 = ops.pack_weight((torch.randn(128, 3, 3, 3, 128, generator=g) / np.sqrt(27 * 128)).to(dev), 0)
    for flip in (False, True):
        ref = ops.conv_gemm(x2, w2, 128, 27, rb.pair, rb.ld, n, flip_k=flip)
        out = ops.conv_gemm_balanced(x2, w2, 128, 27, rb.pair, rb.ld, n, plan, flip_k=flip)
        assert _rel_t(out, ref) < 2e-6


def _rel_t(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))


# ------------------------------------------------------------------------------------------ voxel query (row f-4)

@pytest.mark.parametrize("nsample,ranges,radius", [(16, (1, 1, 1), 0.4), (4, (2, 3, 3), 1.0), (32, (1, 4, 4), 5.0)])
def test_voxel_query_matches_oracle(nsample, ranges, radius, orc):
    """spx_voxel_query against the oracle restatement of voxel_query_kernel_stack on the conv3-level voxels of two
    synthetic frames: index lists (including the fork's reservoir replacement once a ball overflows), empty-ball marker
    and occupied-cell counts — exact.  (The reservoir's random stream itself is parity-unpinned, see csrc/voxel_query.hip.)"""
    import spx
    from pcdet_amd.ops.pointnet2.pointnet2_stack import voxel_query_utils as vq
    from pcdet_amd.utils import common_utils
    from spx import ops
    idx_np, shape = _frame_indices(orc, 0, 2)
    dev = _dev()
    st = spx.SparseConvTensor(torch.zeros(idx_np.shape[0], 1, device=dev), torch.from_numpy(idx_np).to(dev), shape, 2)
    table = common_utils.generate_voxel2pinds(st)
    centers = common_utils.get_voxel_centers(st.indices[:, 1:], 1, [0.05, 0.05, 0.1], [0, -8, -3, 12.8, 8, 1]).contiguous()
    g = torch.Generator().manual_seed(5)
    pick = torch.randperm(idx_np.shape[0], generator=g)[:1500].to(dev)
    q_xyz = (centers[pick] + 0.01).contiguous()
    q_coords = st.indices[pick].clone()
    q_coords[:7, 1:] = torch.tensor([0, 0, 0], dtype=torch.int32, device=dev)        # corner queries: clipped scan
    idx, cnt = ops.voxel_query(q_xyz, centers, q_coords, table, nsample, radius, ranges)
    idx_o, cnt_o = orc.voxel_query(q_xyz.cpu().numpy(), centers.cpu().numpy(), q_coords.cpu().numpy(), table.cpu().numpy(),
                                   nsample, radius, ranges)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o)
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert (cnt_o > 0).any() and ((idx_o[:, 0] == -1).any() or radius > 1)
    # python wrapper: empty balls -> zeros + mask, density in (0, 1]
    i2, empty, dens = vq.voxel_query(ranges, radius, nsample, centers, q_xyz, q_coords, table)
    assert torch.equal(empty.cpu(), torch.from_numpy(idx_o[:, 0] == -1)) and int(i2.min()) >= 0
    vol = (2 * ranges[0] + 1) * (2 * ranges[1] + 1) * (2 * ranges[2] + 1)
    assert torch.allclose(dens.cpu().view(-1), torch.from_numpy(cnt_o).float() / vol)
    # grouping: features gathered through the lists, frame-local indexing
    feats = torch.randn(idx_np.shape[0], 8, generator=g).to(dev)
    nb = torch.bincount(st.indices[:, 0].long(), minlength=2)
    order = torch.argsort(q_coords[:, 0].long(), stable=True)                        # queries grouped by frame
    mod = vq.VoxelQueryAndGrouping(ranges, radius, nsample)
    gf, gx, em, _ = mod(q_coords[order].contiguous(), centers, nb, q_xyz[order].contiguous(),
                        torch.bincount(q_coords[:, 0].long(), minlength=2), feats, table)
    assert gf.shape == (1500, 8, nsample) and gx.shape == (1500, 3, nsample)
    idx_r, _ = orc.voxel_query(q_xyz[order].cpu().numpy(), centers.cpu().numpy(), q_coords[order].cpu().numpy(),
                               table.cpu().numpy(), nsample, radius, ranges)     # the reservoir is seeded by the query's position
    ref_rows = torch.from_numpy(idx_r).to(dev).long()
    ref_rows[em] = 0
    nonempty = ~em
    assert torch.equal(gf[nonempty], feats[ref_rows[nonempty]].permute(0, 2, 1))


@pytest.mark.parametrize("nsample,ranges,strides,former,radius", [(16, (2, 4, 4), (1, 2, 2), 0.2, 0.9),
                                                                    (3, (2, 6, 6), (2, 3, 3), 0.0, 5.0),
                                                                    (4, (1, 2, 2), (1, 1, 1), 0.0, 0.6)])
def test_voxel_query_dilated_matches_oracle(nsample, ranges, strides, former, radius, orc):
    """spx_voxel_query_dilated against the oracle restatement of voxel_query_dilated_kernel_stack: strided scan, the
    inner-radius cut, filled-slot counts, padding and the overflow reservoir — exact.  Stride 1 with no inner radius
    must give the plain kernel's lists."""
    import spx
    from pcdet_amd.ops.pointnet2.pointnet2_stack import voxel_query_utils as vq
    from pcdet_amd.utils import common_utils
    from spx import ops
    idx_np, shape = _frame_indices(orc, 0, 2)
    dev = _dev()
    st = spx.SparseConvTensor(torch.zeros(idx_np.shape[0], 1, device=dev), torch.from_numpy(idx_np).to(dev), shape, 2)
    table = common_utils.generate_voxel2pinds(st)
    centers = common_utils.get_voxel_centers(st.indices[:, 1:], 1, [0.05, 0.05, 0.1], [0, -8, -3, 12.8, 8, 1]).contiguous()
    g = torch.Generator().manual_seed(9)
    pick = torch.sort(torch.randperm(idx_np.shape[0], generator=g)[:1200]).values.to(dev)   # batch-major like the rows
    q_xyz = (centers[pick] + 0.01).contiguous()
    q_coords = st.indices[pick].clone()
    idx, cnt, filled = ops.voxel_query_dilated(q_xyz, centers, q_coords, table, nsample, former, radius, ranges, strides)
    idx_o, cnt_o, filled_o = orc.voxel_query_dilated(q_xyz.cpu().numpy(), centers.cpu().numpy(), q_coords.cpu().numpy(),
                                                     table.cpu().numpy(), nsample, former, radius, ranges, strides)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o) and np.array_equal(filled.cpu().numpy(), filled_o)
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert filled_o.max() == nsample or radius < 1                                    # the reservoir branch ran
    if former > 0:
        assert (idx_o[:, 0] != pick.cpu().numpy()).all()                             # the query's own voxel is cut
    if strides == (1, 1, 1) and former == 0.0:
        i_plain, c_plain = ops.voxel_query(q_xyz, centers, q_coords, table, nsample, radius, ranges)
        assert torch.equal(i_plain, idx) and torch.equal(c_plain, cnt)
    i2, empty, score = vq.voxel_query_dilated(ranges, strides, former, radius, nsample, centers, q_xyz, q_coords, table)
    assert torch.equal(empty.cpu(), torch.from_numpy(idx_o[:, 0] == -1)) and int(i2.min()) >= 0
    assert torch.allclose(score.cpu().view(-1), torch.clamp(torch.from_numpy(cnt_o).float() / nsample, max=1.0))
    feats = torch.randn(idx_np.shape[0], 6, generator=g).to(dev)
    nb = torch.bincount(st.indices[:, 0].long(), minlength=2)
    mod = vq.VoxelQueryAndGroupingDilated(ranges, strides, former, radius, nsample)
    gf, gx, em, _ = mod(q_coords, centers, nb, q_xyz, torch.bincount(q_coords[:, 0].long(), minlength=2), feats, table)
    rows = torch.from_numpy(idx_o).to(dev).long()
    rows[em] = 0
    assert torch.equal(gf[~em], feats[rows[~em]].permute(0, 2, 1)) and torch.equal(gx[~em], centers[rows[~em]].permute(0, 2, 1))


# ------------------------------------------------------------------------------------------ densify

@pytest.mark.parametrize("channels_last", [False, True])
def test_densify_and_backward(channels_last, orc):
    from spx import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    batch, shape, c = 3, [2, 20, 18], 128
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = torch.randperm(cells, generator=g)[:500].sort()[0]
    vol = shape[0] * shape[1] * shape[2]
    idx = torch.stack([lin // vol, (lin % vol) // (shape[1] * shape[2]), (lin // shape[2]) % shape[1],
                       lin % shape[2]], 1).int()
    feat = torch.randn(500, c, generator=g)
    dense = ops.densify(feat.to(dev), idx.to(dev), batch, shape, channels_last=channels_last)
    assert list(dense.shape) == [batch, c, *shape]
    ref = orc.densify(feat.numpy(), idx.numpy(), batch, shape)
    assert np.array_equal(dense.cpu().numpy(), ref)
    # height_compression.py:22-23: view(N, C*D, H, W) must be legal on what dense() returns
    bev = dense.view(batch, c * shape[0], shape[1], shape[2])
    assert np.array_equal(bev.cpu().numpy(), ref.reshape(batch, c * shape[0], shape[1], shape[2]))
    if channels_last:
        assert bev.is_contiguous(memory_format=torch.channels_last)
    dd = torch.randn(batch, c, *shape, generator=g)
    dfeat = ops.densify_bwd(dd.to(dev), idx.to(dev), batch, shape, channels_last=channels_last)
    ii = idx.long()
    assert torch.equal(dfeat.cpu(), dd[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]])


# ------------------------------------------------------------------------------------------ voxelise

def _vox_check(ops, orc, frames, geom_range, vsize, max_points, max_voxels):
    """frames: list of [Np, C] arrays -> GPU batched voxelise vs per-frame oracle + concat (dataset.py:161-229)."""
    dev = _dev()
    pts = np.concatenate([np.concatenate([np.full((f.shape[0], 1), b, np.float32), f], 1)
                          for b, f in enumerate(frames)], 0)
    out = ops.voxelize(torch.from_numpy(pts).to(dev), geom_range, vsize, max_points, max_voxels,
                       batch_size=len(frames), batch_col=0, xyz_col=1, feat_col=1)
    vs, cs, ns = [], [], []
    for b, f in enumerate(frames):
        v, c, n = orc.voxelize(f, geom_range, vsize, max_points, max_voxels)
        vs.append(v)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
        ns.append(n)
    v, c, n = np.concatenate(vs), np.concatenate(cs), np.concatenate(ns)
    assert out["num_voxels"] == v.shape[0]
    assert np.array_equal(out["coords"].cpu().numpy(), c)
    assert np.array_equal(out["num_points"].cpu().numpy(), n)
    assert np.array_equal(out["voxels"].cpu().numpy(), v)
    _close(out["mean"].cpu().numpy(), orc.mean_vfe(v, n), tol=1e-6)
    return out


def test_voxelize_synthetic_frames(orc):
    from pcdet_amd.datasets import synthetic as syn
    from spx import ops
    g = syn.KITTI
    frames = [syn.make_frame(1, i)["points"] for i in range(3)]
    _vox_check(ops, orc, frames, g["point_cloud_range"], g["voxel_size"], 5, 16000)
    # voxel budget smaller than the scene: first-occurrence voxels survive, later ones (and their points) drop
    _vox_check(ops, orc, frames, g["point_cloud_range"], g["voxel_size"], 5, 3000)
    _vox_check(ops, orc, frames, g["point_cloud_range"], g["voxel_size"], 2, 4999)
    gw = syn.WAYMO
    frames = [syn.make_frame(3, i)["points"] for i in range(2)]
    _vox_check(ops, orc, frames, gw["point_cloud_range"], gw["voxel_size"], 5, 150000)


def test_voxelize_known_answers(orc):
    """Hand-built points hitting each rule: range edges (hi - 0.0002, common_utils.py:66-70), exact cell
    boundaries, negative / out-of-range / NaN coordinates, > max_points in a voxel, empty frame."""
    from spx import ops
    rng = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]
    vs = [0.05, 0.05, 0.1]
    r = np.random.default_rng(5)
    pts = [
        [0.0, -40.0, -3.0, 0.1], [70.4 - 0.0002, 40.0 - 0.0002, 1.0 - 0.0002, 0.2], [70.4, 0.0, 0.0, 0.3],
        [-1e-6, 0.0, 0.0, 0.4], [35.0, 40.0, 0.0, 0.5], [0.05, 0.05, 0.1, 0.6], [0.0499999, 0.0, 0.0, 0.7],
        [np.nan, 0.0, 0.0, 0.8], [10.0, np.inf, 0.0, 0.9], [0.15, -39.95, -2.9, 1.0],
    ]
    crowd = np.concatenate([np.full((9, 1), 12.34), np.full((9, 1), 5.67), np.full((9, 1), -1.0),
                            np.arange(9)[:, None] / 10.0], 1) + np.concatenate([r.random((9, 3)) * 0.01,
                                                                               np.zeros((9, 1))], 1)
    bounds = np.stack([np.arange(200) * 0.05, np.arange(200) * 0.05 - 5.0, np.arange(200) % 40 * 0.1 - 3.0,
                       np.arange(200) / 200.0], 1)
    f0 = np.concatenate([np.asarray(pts), crowd, bounds], 0).astype(np.float32)
    f1 = np.zeros((0, 4), np.float32)           # empty frame in the middle of a batch
    f2 = f0[::-1].copy()                        # same points, reversed order -> different voxel numbering
    out = _vox_check(ops, orc, [f0, f1, f2], rng, vs, 5, 16000)
    assert int(out["num_points"].max()) == 5
    _vox_check(ops, orc, [f0, f1, f2], rng, vs, 3, 50)
    _vox_check(ops, orc, [f1], rng, vs, 5, 10)


def test_mean_vfe_standalone(orc):
    from spx import ops
    g = torch.Generator().manual_seed(11)
    num = torch.randint(0, 6, (1000,), generator=g).int()
    vox = torch.randn(1000, 5, 4, generator=g) * (torch.arange(5)[None, :, None] < num[:, None, None])
    out = ops.mean_vfe(vox.to(_dev()), num.to(_dev()))
    _close(out.cpu().numpy(), orc.mean_vfe(vox.numpy(), num.numpy()), tol=1e-6)


# ------------------------------------------------------------------------------------------ rotated BEV IoU / NMS (row f-1)

def _rand_boxes(g, n, spread, cluster=False):
    if cluster:
        centres = torch.rand(max(n // 40, 1), 2, generator=g) * spread
        xy = centres[torch.randint(0, centres.shape[0], (n,), generator=g)] + torch.randn(n, 2, generator=g) * 0.7
    else:
        xy = torch.rand(n, 2, generator=g) * spread
    dims = torch.rand(n, 2, generator=g) * torch.tensor([3.5, 1.5]) + torch.tensor([0.6, 0.5])
    return torch.cat([xy, torch.zeros(n, 1), dims, torch.ones(n, 1) * 1.5, (torch.rand(n, 1, generator=g) - 0.5) * 6.3], 1)


def test_boxes_iou_bev_vs_oracle(orc):
    from spx import ops
    g = torch.Generator().manual_seed(21)
    a = _rand_boxes(g, 300, 30.0, cluster=True)
    b = a[:257].clone()
    b[:, :2] += torch.randn(257, 2, generator=g) * 0.8      # overlapping neighbours of a
    b[:, 6] += torch.randn(257, generator=g)
    for overlap_only in (False, True):
        out = ops.boxes_iou_bev(a.to(_dev()), b.to(_dev()), overlap_only=overlap_only).cpu().numpy()
        ref = orc.boxes_iou_bev(a.numpy(), b.numpy(), overlap_only)
        assert (ref > 0).mean() > 0.005
        assert np.abs(out - ref).max() < 2e-5
    # identical, contained and axis-aligned special cases (closed forms checked in tests/test_oracle_golden.py)
    same = ops.boxes_iou_bev(a[:50].to(_dev()), a[:50].to(_dev())).diagonal().cpu().numpy()
    assert np.abs(same - 1.0).max() < 1e-4


@pytest.mark.parametrize("n,cluster,thresh,normal", [(1, False, 0.1, False), (64, True, 0.1, False), (65, True, 0.01, False),
                                                     (1000, False, 0.01, False), (3000, True, 0.01, False),
                                                     (4096, True, 0.5, False), (5000, True, 0.3, False),
                                                     (2000, True, 0.2, True)])
def test_nms_bev_vs_oracle(n, cluster, thresh, normal, orc):
    """Greedy NMS keep list (integer result) must equal the oracle's exactly."""
    from spx import ops
    g = torch.Generator().manual_seed(1000 + n)
    boxes = _rand_boxes(g, n, 70.0 if not cluster else 40.0, cluster)
    keep, cnt = ops.nms_bev(boxes.to(_dev()), thresh, axis_aligned=normal)
    got = keep[:int(cnt.item())].cpu().numpy()
    ref = orc.nms_bev(boxes.numpy(), thresh, normal)
    assert 0 < ref.shape[0] <= n
    assert np.array_equal(got, ref)


def test_nms_utils_api(orc):
    """iou3d_nms_utils.nms_gpu signature: unsorted scores in, indices into the ORIGINAL order out."""
    from pcdet_amd.ops.iou3d_nms import iou3d_nms_utils
    g = torch.Generator().manual_seed(77)
    boxes = _rand_boxes(g, 1500, 40.0, True)
    scores = torch.rand(1500, generator=g)
    sel, _ = iou3d_nms_utils.nms_gpu(boxes.to(_dev()), scores.to(_dev()), 0.1, pre_maxsize=1024)
    order = scores.sort(0, descending=True)[1][:1024]
    ref = order[torch.from_numpy(orc.nms_bev(boxes[order].numpy(), 0.1))]
    assert torch.equal(sel.cpu(), ref)
    iou3d = iou3d_nms_utils.boxes_iou3d_gpu(boxes[:20].to(_dev()), boxes[:20].to(_dev()))
    assert float((iou3d.diagonal() - 1).abs().max()) < 1e-4


# ------------------------------------------------------------------------------------------ BatchNorm1d + ReLU (row a10)

@pytest.mark.parametrize("n,c,relu", [(5000, 16, True), (83083, 32, True), (40001, 64, False), (777, 128, True), (2, 64, True)])
def test_fused_bn_relu_train_matches_torch(n, c, relu):
    """csrc/bn_relu.hip against nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU in training mode: output, running
    statistics, and all three gradients."""
    import spx
    from spx import functional as F_
    dev = _dev()
    k = 1.0 if n > 16 else 50.0   # two rows: 1/std of a 2-sample variance amplifies single ulps
    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, c, generator=g) * torch.linspace(0.5, 3.0, c) + torch.linspace(-2, 2, c)).to(dev)
    dy = torch.randn(n, c, generator=g).to(dev)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
    with torch.no_grad():
        ref.weight.copy_(torch.rand(c, generator=g) + 0.5)
        ref.bias.copy_(torch.randn(c, generator=g) * 0.2)
    import copy
    mine = copy.deepcopy(ref)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy)
    xm = x.clone().requires_grad_(True)
    ym = F_.bn_relu_train(xm, mine, relu)
    ym.backward(dy)
    _close(ym.detach().cpu().numpy(), yr.detach().cpu().numpy(), tol=2e-5 * k)
    _close(mine.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), tol=1e-6)
    _close(mine.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), tol=1e-6)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    _close(xm.grad.cpu().numpy(), xr.grad.cpu().numpy(), tol=5e-5 * k)
    _close(mine.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), tol=5e-5 * k)
    _close(mine.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), tol=5e-5 * k)
    # bitwise reproducible
    xm2 = x.clone().requires_grad_(True)
    assert torch.equal(F_.bn_relu_train(xm2, copy.deepcopy(ref), relu), ym)


@pytest.mark.parametrize("n,c,relu", [(83083, 32, True), (40001, 64, True), (5000, 128, False)])
def test_fused_bn_add_relu_train_matches_torch(n, c, relu):
    """spx_bn_add_relu_* (the tail of SparseBasicBlock, reference spconv_backbone.py:56-72: bn2, + identity, ReLU) against
    the torch modules in that order: output, running statistics, and the gradients of x, the identity branch, gamma and
    beta.  The identity values are nudged away from exact cancellation so that both sides take the same ReLU branch."""
    import copy
    from spx import functional as F_
    dev = _dev()
    g = torch.Generator().manual_seed(3 * n + c)
    x = (torch.randn(n, c, generator=g) * torch.linspace(0.5, 3.0, c) + torch.linspace(-2, 2, c)).to(dev)
    res = torch.randn(n, c, generator=g).to(dev)
    dy = torch.randn(n, c, generator=g).to(dev)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
    with torch.no_grad():
        ref.weight.copy_(torch.rand(c, generator=g) + 0.5)
        ref.bias.copy_(torch.randn(c, generator=g) * 0.2)
        z0 = copy.deepcopy(ref)(x) + res
        res = torch.where(z0.abs() < 1e-3, res + 0.01, res)
    mine = copy.deepcopy(ref)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    yr = ref(xr) + rr
    yr = torch.relu(yr) if relu else yr
    yr.backward(dy)
    xm, rm = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    ym = F_.bn_act(xm, mine, relu, rm)
    ym.backward(dy)
    _close(ym.detach().cpu().numpy(), yr.detach().cpu().numpy(), tol=2e-5)
    _close(mine.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), tol=1e-6)
    _close(mine.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), tol=1e-6)
    assert torch.equal(rm.grad, rr.grad)                       # dy masked by the ReLU: no arithmetic, exact
    _close(xm.grad.cpu().numpy(), xr.grad.cpu().numpy(), tol=5e-5)
    _close(mine.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), tol=5e-5)
    _close(mine.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), tol=5e-5)
    # eval mode / CPU-side fallback of bn_act is the torch modules themselves
    mine.eval()
    with torch.no_grad():
        assert torch.equal(F_.bn_act(x, mine, relu, res), torch.relu(mine(x) + res) if relu else mine(x) + res)


def test_fused_bn_relu_cat_matches_torch():
    """bn_relu_cat_train (BN+ReLU of several branches written into the channel slices of one matrix, gradients read from
    the slices in place; the BEV up-sampling concatenation, reference base_bev_backbone.py:99-106) against
    torch.cat([relu(bn_i(x_i))], 1) with nn.BatchNorm1d: output, all gradients, running statistics, counters."""
    import copy
    from spx import functional as F_
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    n, widths = 30011, (256, 128, 64)
    xs = [(torch.randn(n, c, generator=g) * 1.7 + 0.3).to(dev) for c in widths]
    refs = [torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train() for c in widths]
    with torch.no_grad():
        for r in refs:
            r.weight.copy_(torch.rand(r.num_features, generator=g) + 0.5)
            r.bias.copy_(torch.randn(r.num_features, generator=g) * 0.2)
        for _ in range(3):          # keep pre-activations away from 0 so that both sides take the same ReLU branch
            for r, x in zip(refs, xs):
                z = copy.deepcopy(r)(x)
                x[z.abs() < 1e-3] += 0.02
        assert min(float(copy.deepcopy(r)(x).abs().min()) for r, x in zip(refs, xs)) > 1e-5
    mine = copy.deepcopy(refs)
    dy = torch.randn(n, sum(widths), generator=g).to(dev)
    xr = [x.clone().requires_grad_(True) for x in xs]
    yr = torch.cat([torch.relu(r(x)) for r, x in zip(refs, xr)], 1)
    yr.backward(dy)
    xm = [x.clone().requires_grad_(True) for x in xs]
    ym = F_.bn_relu_cat_train(xm, mine)
    assert ym.shape == yr.shape and ym.is_contiguous()
    ym.backward(dy)
    _close(ym.detach().cpu().numpy(), yr.detach().cpu().numpy(), tol=2e-5)
    for a, b, m, r in zip(xm, xr, mine, refs):
        _close(a.grad.cpu().numpy(), b.grad.cpu().numpy(), tol=5e-5)
        _close(m.weight.grad.cpu().numpy(), r.weight.grad.cpu().numpy(), tol=5e-5)
        _close(m.bias.grad.cpu().numpy(), r.bias.grad.cpu().numpy(), tol=5e-5)
        _close(m.running_mean.cpu().numpy(), r.running_mean.cpu().numpy(), tol=1e-6)
        _close(m.running_var.cpu().numpy(), r.running_var.cpu().numpy(), tol=1e-6)
        assert int(m.num_batches_tracked) == int(r.num_batches_tracked) == 1
    # a gradient that arrives as a non-contiguous view takes the copy route and gives the same numbers
    xm2 = [x.clone().requires_grad_(True) for x in xs]
    ym2 = F_.bn_relu_cat_train(xm2, copy.deepcopy(refs))
    wide = torch.zeros(n, sum(widths) + 3, device=dev)
    wide[:, 3:] = dy
    ym2.backward(wide[:, 3:])
    for a, b in zip(xm2, xm):
        assert torch.equal(a.grad, b.grad)


# ------------------------------------------------------------------------------------------ bounded hash probes

def test_dirty_precleared_workspace_reports_table_full_instead_of_hanging(orc):
    """SPX_WS_PRECLEARED on a workspace that is NOT clean (every hash key slot holds a foreign key): the voxeliser and the
    submanifold rulebook must come back with SPX_ERR_TABLE_FULL in the status word — every probe loop is bounded by the
    slot count (round 1 recorded an endless CAS probe on a stale workspace) — and the next ordinary call must be exact."""
    from spx import _lib, ops
    from pcdet_amd.datasets import synthetic as syn
    dev = _dev()
    geom = syn.KITTI
    frame = syn.make_frame(1, 0)
    pts = torch.from_numpy(frame["points"][:4096]).to(dev)
    lib = _lib.load()
    wsb = max(lib.spx_voxelize_ws_bytes(pts.shape[0], 1, 5), lib.spx_subm_rulebook_ws_bytes(4096))
    ws = ops.workspace(dev, wsb)
    ws.view(torch.int32).fill_(0x01010101)          # no slot is EMPTY (0xFF..), no key of this frame matches
    with pytest.raises(_lib.SpxError, match="hash table full"):
        ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, 16000, ws_precleared=True)
    vox = ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, 16000)      # library clears: exact again
    v_o, c_o, _n = orc.voxelize(frame["points"][:4096], geom["point_cloud_range"], geom["voxel_size"], 5, 16000)
    assert np.array_equal(vox["coords"][:, 1:].cpu().numpy(), c_o)
    assert np.array_equal(vox["voxels"].cpu().numpy(), v_o)

    idx = vox["coords"]
    shape = [41, 1600, 1408]
    ops.check_status(dev)                           # clean so far
    ops.workspace(dev, wsb).view(torch.int32).fill_(0x01010101)
    rb = ops.subm_rulebook(idx, 1, shape, (3, 3, 3), ws_precleared=True)
    torch.cuda.synchronize()                        # returns: bounded probes
    with pytest.raises(_lib.SpxError, match="hash table full"):
        ops.check_status(dev)
    ops.check_status(dev)                           # the word was reset by the failed check
    assert int((rb.pair >= 0).sum()) == 0           # nothing was found in the foreign table
    rb = ops.subm_rulebook(idx, 1, shape, (3, 3, 3))
    pair_o, _ = orc.subm_rulebook(idx.cpu().numpy(), shape)
    assert np.array_equal(rb.pair[:, :rb.n_out].cpu().numpy(), pair_o)
    ops.check_status(dev)
    assert lib.spx_read_status(ctypes_ptr(ops.status_word(dev)), None) == 0


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def test_conv_balanced_empty_tiles_and_combine(orc):
    """Rule tables with whole 64-row super-tiles that have no pair at all (they must come out as epilogue(0), written by
    the main kernel now that the fix-up pass is gone), a table with no pair anywhere, and the in-launch combine of
    super-tiles split between workgroups: against spx_conv_gemm on the same tables, bitwise stable over repeated launches
    of the same plan (the arrival counters are reset by each last arriver)."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 2, 2)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    g = torch.Generator().manual_seed(23)
    sub = ops.subm_rulebook(d_idx, 2, shape, (3, 3, 3))
    n = sub.n_out
    holes = sub.pair.clone()
    holes[:, 128:320] = -1                       # three empty super-tiles in the middle
    holes[:, n - 100:] = -1                      # and a ragged empty tail
    none = torch.full_like(sub.pair, -1)
    for (cs, cd) in ((64, 64), (128, 128)):
        w = (torch.randn(cd, 3, 3, 3, cs, generator=g) / np.sqrt(27 * cs)).to(dev)
        wp = ops.pack_weight(w, 0)
        x = torch.randn(n, cs, generator=g).to(dev)
        sh = torch.randn(cd, generator=g).to(dev)
        for pair in (holes, none, sub.pair):
            plan = ops.conv_plan(pair, sub.ld, 27, n)
            for relu in (False, True):
                ref = ops.conv_gemm(x, wp, cd, 27, pair, sub.ld, n, shift=sh, relu=relu)
                outs = [ops.conv_gemm_balanced(x, wp, cd, 27, pair, sub.ld, n, plan, shift=sh, relu=relu) for _ in range(3)]
                assert _rel_t(outs[0], ref) < 2e-6
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
        # rows of an empty super-tile are exactly epilogue(0)
        plan = ops.conv_plan(holes, sub.ld, 27, n)
        out = ops.conv_gemm_balanced(x, wp, cd, 27, holes, sub.ld, n, plan, shift=sh, relu=True)
        assert torch.equal(out[128:320], torch.relu(sh).expand(192, cd))


# ------------------------------------------------------------------------------------------ strided + submanifold tables in one build

@pytest.mark.parametrize("cfg_id,nframes", [(1, 2), (2, 3)])
def test_level_build_strided_plus_subm_from_bitmap(cfg_id, nframes, orc):
    """spx_conv_rulebook with subm_ksize: the strided tables AND the submanifold table of the output level from one rank
    bitmap (no hash), chained through the four levels of VoxelBackBone8x: everything bit-exact against the C oracle and
    identical to the hash-built submanifold table; exact-size and static-capacity (device-side counts) forms."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, cfg_id, nframes)
    dev = _dev()
    chain = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    for k, s, p in chain:
        d_idx = torch.from_numpy(idx_np).to(dev)
        rb = ops.conv_rulebook(d_idx, nframes, shape, k, s, p, want_cnt=True, subm_ksize=(3, 3, 3))
        oi, pf, pb, cnt_o, oshape = orc.conv_rulebook(idx_np, shape, k, s, p)
        assert rb.n_out == oi.shape[0] and rb.out_shape == oshape
        assert np.array_equal(rb.out_indices.cpu().numpy(), oi)
        assert np.array_equal(rb.pair[:, :rb.n_out].cpu().numpy(), pf)
        assert np.array_equal(rb.pair_bwd.cpu().numpy(), pb)
        assert np.array_equal(rb.cnt.cpu().numpy(), cnt_o)
        sub = rb.subm_next
        pair_o, scnt_o = orc.subm_rulebook(oi, oshape, (3, 3, 3))
        assert sub.subm and sub.n_out == rb.n_out and sub.out_indices is rb.out_indices
        assert np.array_equal(sub.pair[:, :sub.n_out].cpu().numpy(), pair_o)
        assert np.array_equal(sub.cnt.cpu().numpy(), scnt_o)
        hashed = ops.subm_rulebook(rb.out_indices, nframes, oshape, (3, 3, 3))
        assert torch.equal(hashed.pair[:, :rb.n_out], sub.pair[:, :rb.n_out])
        # static capacity: input at capacity with a device-side live count, output capped above the live count
        n = idx_np.shape[0]
        pad = torch.cat([d_idx, torch.full((257, 4), 7, dtype=torch.int32, device=dev)], 0)      # garbage rows beyond the live count
        d_n = torch.tensor([n], dtype=torch.int64, device=dev)
        rs = ops.conv_rulebook(pad, nframes, shape, k, s, p, d_n_in=d_n, cap=rb.n_out + 100, sync=False, subm_ksize=(3, 3, 3))
        assert int(rs.d_n_out) == rb.n_out
        assert np.array_equal(rs.out_indices[:rb.n_out].cpu().numpy(), oi)
        assert np.array_equal(rs.pair[:, :rb.n_out].cpu().numpy(), pf)
        assert np.array_equal(rs.pair_bwd[:, :n].cpu().numpy(), pb)
        assert np.array_equal(rs.subm_next.pair[:, :rb.n_out].cpu().numpy(), pair_o)
        idx_np, shape = oi, oshape


def test_level_build_odd_strides_and_kernels(orc):
    """Strides that are not powers of two (the integer-division form of the candidate test), anisotropic kernels,
    dilation, a submanifold kernel other than 3x3x3, an empty input."""
    from spx import ops
    dev = _dev()
    rs = np.random.RandomState(4)
    shape = [13, 17, 19]
    lin = np.sort(rs.permutation(2 * 13 * 17 * 19)[:1500])
    idx_np = np.stack([lin // (13 * 17 * 19), (lin // (17 * 19)) % 13, (lin // 19) % 17, lin % 19], 1).astype(np.int32)
    d_idx = torch.from_numpy(idx_np).to(dev)
    for k, s, p, d, ks in (((3, 3, 3), (3, 3, 3), (1, 1, 1), (1, 1, 1), (3, 3, 3)), ((3, 2, 1), (1, 2, 3), (1, 0, 0), (1, 1, 1), (1, 3, 3)),
                           ((2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), (3, 1, 1)), ((3, 3, 3), (2, 3, 1), (2, 2, 2), (2, 2, 2), (3, 3, 3))):
        rb = ops.conv_rulebook(d_idx, 2, shape, k, s, p, d, want_cnt=True, subm_ksize=ks)
        oi, pf, pb, cnt_o, oshape = orc.conv_rulebook(idx_np, shape, k, s, p, d)
        assert np.array_equal(rb.out_indices.cpu().numpy(), oi) and rb.out_shape == oshape
        assert np.array_equal(rb.pair[:, :rb.n_out].cpu().numpy(), pf)
        assert np.array_equal(rb.pair_bwd.cpu().numpy(), pb)
        assert np.array_equal(rb.cnt.cpu().numpy(), cnt_o)
        pair_o, scnt_o = orc.subm_rulebook(oi, oshape, ks)
        assert np.array_equal(rb.subm_next.pair[:, :rb.n_out].cpu().numpy(), pair_o)
        assert np.array_equal(rb.subm_next.cnt.cpu().numpy(), scnt_o)
    e = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    rb = ops.conv_rulebook(e, 1, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), subm_ksize=(3, 3, 3))
    assert rb.n_out == 0 and rb.subm_next.n_out == 0


def test_pack_all_matches_per_weight_pack():
    """spx_pack_weight_batched (every conv weight of a backbone, both operand orders, one launch) lays down exactly what
    spx_pack_weight mode 2 does per weight — MFMA order for 16/32/64/128-channel pairs, plain order for the 4-channel
    input conv — and refreshes the per-parameter caches only when a weight changed."""
    from pcdet_amd.config import AttrDict
    from pcdet_amd.models.backbones_3d import VoxelBackBone8x
    import spx
    from spx import functional as F_, ops
    dev = _dev()
    torch.manual_seed(2)
    net = VoxelBackBone8x(AttrDict(), 4, [32, 32, 8]).to(dev)
    convs = [m for m in net.modules() if isinstance(m, spx.SparseConvolution)]
    assert len(convs) == 12
    assert F_.pack_all(convs) is True
    for m in convs:
        both = ops.pack_weight(m.weight, 2)
        half = both.numel() // 2
        assert torch.equal(F_._packed(m.weight, 0), both[:half]) and torch.equal(F_._packed(m.weight, 1), both[half:])
    assert F_.pack_all(convs) is False                       # nothing changed: no launch
    with torch.no_grad():
        convs[3].weight.mul_(1.5)                            # an in-place update (optimizer step) invalidates
    assert F_.pack_all(convs) is True
    both = ops.pack_weight(convs[3].weight, 2)
    assert torch.equal(F_._packed(convs[3].weight, 0), both[:both.numel() // 2])
