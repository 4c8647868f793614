"""Round-3 schedule of the sparse convolution (csrc/conv_ring.hip -> spx_conv_ring_plan / spx_conv_gemm_ring), through the
C ABI: every case against spx_conv_gemm on the same tables (which test_gpu_kernels.py pins to the oracle and the dense-conv
goldens), so a pass here means oracle parity to the same bar.  Both kernels sum every output row over k in ascending order
with the same fp32 MFMA chain: results are expected to be BIT-IDENTICAL, whatever the plan.

Reference call sites: pcdet/models/backbones_3d/spconv_backbone.py:105-114 (the 64-channel submanifold / strided layers)
and their autograd.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True, params=[0, 1, 2], ids=["by_size", "one_tile_per_wave", "two_tiles_per_wave"])
def tiles_per_wave(request):
    """Both decompositions of the kernel (ring_body1 / ring_body2), forced, and the plan's own choice by size."""
    from spx import _lib
    lib = _lib.load()
    default = lib.spx_conv_ring_tiles_per_wave(-1)
    assert lib.spx_conv_ring_tiles_per_wave(request.param) == request.param
    yield request.param
    lib.spx_conv_ring_tiles_per_wave(default)


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


def _plan_header(plan):
    h = plan[:32].cpu().numpy()
    return dict(tiles=int(h[0]), units=int(h[1]), timeouts=int(h[2]), chunks=int(h[3]), tm=int(h[4]), rounds=h[8:16].tolist(),
                nchunks=h[16:24].tolist())


def _check_plan(ops, plan, pair, ld, K, n, live):
    """Every live tile is dealt exactly once, with its offset mask; no spin ever gave up."""
    from spx import _lib
    hdr = _plan_header(plan)
    T = (live + 15) // 16
    assert hdr["tiles"] == T and hdr["timeouts"] == 0
    m = (pair[:K, :live] >= 0)
    pad = (-live) % 16
    if pad:
        m = torch.cat([m, torch.zeros((K, pad), dtype=torch.bool, device=m.device)], 1)
    tile_mask = (m.view(K, T, 16).any(2).to(torch.int64) << torch.arange(K, device=m.device)[:, None]).sum(0)
    assert hdr["units"] == int(m.view(K, T, 16).any(2).sum())
    lib = _lib.load()
    nbytes = lib.spx_conv_ring_plan_bytes(n)
    assert plan.numel() * 4 == nbytes
    tcap = (n + 15) // 16 + 1
    kHdr = 32 + 8 * 96
    off_sorted = kHdr + (tcap + 3) // 4 * 4
    off_ent = off_sorted + (tcap + 3) // 4 * 4
    R, tm = max(hdr["rounds"]), hdr["tm"]
    want = lib.spx_conv_ring_tiles_per_wave(-1)
    assert tm == (want if want else (2 if (T + 7) // 8 > 384 else 1)) or (want == 0 and tm in (1, 2) and abs((T + 7) // 8 - 384) < 100)
    assert R <= 1 or tm == 2 or want == 1
    ent = plan[off_ent: off_ent + R * 256 * 12 * tm * 2].view(R, 256, 12 * tm, 2).cpu()
    seen = torch.zeros(T, dtype=torch.int64)
    for x in range(8):
        e = ent[:hdr["rounds"][x], x::8]                   # workgroups with blockIdx & 7 == x
        tiles = e[..., 0].reshape(-1)
        masks = e[..., 1].reshape(-1)
        ok = tiles >= 0
        seen.index_add_(0, tiles[ok].long(), torch.ones(int(ok.sum()), dtype=torch.int64))
        assert torch.equal(masks[ok].long() & 0xFFFFFFFF, tile_mask.cpu()[tiles[ok].long()] & 0xFFFFFFFF)
    assert int(seen.min()) == 1 and int(seen.max()) == 1
    return hdr


def _rel_t(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))


def test_ring_matches_tile_per_wave_on_real_tables(orc):
    """Forward, flipped (dgrad of a submanifold conv) and strided tables in both directions at the cfg-2 batch size, the four
    channel pairs the kernel is built for, plain and grouped rows, fused epilogue, the BatchNorm statistics of the epilogue;
    bit-identical to spx_conv_gemm and from launch to launch; the plan deals every tile once."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 2, 4)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    g = torch.Generator().manual_seed(31)
    sub = ops.subm_rulebook(d_idx, 4, shape, (3, 3, 3))
    strd = ops.conv_rulebook(d_idx, 4, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    cases = [(sub.pair, sub.ld, sub.n_out, sub.n_in, False), (sub.pair, sub.ld, sub.n_in, sub.n_out, True),
             (strd.pair, strd.ld, strd.n_out, strd.n_in, False),
             (strd.pair_bwd, strd.pair_bwd.shape[1], strd.n_in, strd.n_out, False)]
    plans = [ops.conv_ring_plan(pair, ld, 27, n_dst) for pair, ld, n_dst, _, _ in cases]
    for (pair, ld, n_dst, _, _), plan in zip(cases, plans):
        hdr = _check_plan(ops, plan, pair, ld, 27, n_dst, n_dst)
        assert len(set(hdr["rounds"])) <= 2 and max(hdr["rounds"]) - min(hdr["rounds"]) <= 1
    perm, grouped = ops.conv_group(sub.pair, sub.ld, 27, sub.n_out)
    gplan = ops.conv_ring_plan(grouped, sub.n_out, 27, sub.n_out)
    assert _plan_header(gplan)["units"] < _plan_header(plans[0])["units"]        # grouped rows: fewer MFMA units
    for (cs, cd) in ((64, 64), (32, 64), (64, 32), (32, 32)):
        w = (torch.randn(cd, 3, 3, 3, cs, generator=g) / np.sqrt(27 * cs)).to(dev)
        wp = ops.pack_weight(w, 0)
        for (pair, ld, n_dst, n_src, flip), plan in zip(cases, plans):
            x = torch.randn(n_src, cs, generator=g).to(dev)
            ref = ops.conv_gemm(x, wp, cd, 27, pair, ld, n_dst, flip_k=flip)
            out = ops.conv_gemm_ring(x, wp, cd, 27, pair, ld, n_dst, plan, flip_k=flip)
            assert torch.equal(out, ref)
            assert torch.equal(out, ops.conv_gemm_ring(x, wp, cd, 27, pair, ld, n_dst, plan, flip_k=flip))
        sc = (torch.rand(cd, generator=g) + 0.5).to(dev)
        sh = torch.randn(cd, generator=g).to(dev)
        x = torch.randn(sub.n_in, cs, generator=g).to(dev)
        for flip, kw in ((False, {}), (True, {}), (False, dict(scale=sc, shift=sh, relu=True))):
            ref = ops.conv_gemm(x, wp, cd, 27, sub.pair, sub.ld, sub.n_out, flip_k=flip, **kw)
            out, st = ops.conv_gemm_ring(x, wp, cd, 27, grouped, sub.n_out, sub.n_out, gplan, flip_k=flip, perm=perm,
                                         want_stats=True, **kw)
            assert torch.equal(out, ref)
            s = st.double().sum(0)
            want1, want2 = out.double().sum(0), (out.double() ** 2).sum(0)
            assert float((s[0] - want1).abs().max()) <= 1e-5 * float(out.double().abs().sum(0).max())
            assert float((s[1] - want2).abs().max()) <= 1e-5 * float(want2.max())
    assert all(_plan_header(p)["timeouts"] == 0 for p in plans + [gplan])


@pytest.mark.parametrize("n", [1, 15, 16, 17, 200, 1000, 5000, 6200, 12345])
def test_ring_small_and_ragged_sizes(n):
    """Sizes where most workgroups (or all but one wave) have nothing to do, the last tile is ragged, and (6 200 rows = 388
    tiles... per XCD 48-49) nothing special; 12 345 rows: a partly filled single round."""
    from spx import ops
    dev = _dev()
    g = torch.Generator().manual_seed(100 + n)
    side = max(4, int(np.ceil((2.5 * n) ** (1.0 / 3.0))))
    lin = torch.randperm(side ** 3, generator=g)[:n].sort()[0]
    idx = torch.stack([torch.zeros_like(lin), lin // (side * side), (lin // side) % side, lin % side], 1).int().to(dev)
    rb = ops.subm_rulebook(idx, 1, [side, side, side], (3, 3, 3))
    w = (torch.randn(64, 3, 3, 3, 64, generator=g) / 40.0).to(dev)
    wp = ops.pack_weight(w, 0)
    x = torch.randn(n, 64, generator=g).to(dev)
    plan = ops.conv_ring_plan(rb.pair, rb.ld, 27, n)
    _check_plan(ops, plan, rb.pair, rb.ld, 27, n, n)
    for flip in (False, True):
        ref = ops.conv_gemm(x, wp, 64, 27, rb.pair, rb.ld, n, flip_k=flip)
        out, st = ops.conv_gemm_ring(x, wp, 64, 27, rb.pair, rb.ld, n, plan, flip_k=flip, want_stats=True)
        assert torch.equal(out, ref)
        assert float((st.double().sum(0)[0] - out.double().sum(0)).abs().max()) <= 1e-5 * max(1.0, float(out.abs().sum(0).max()))
    assert _plan_header(plan)["timeouts"] == 0


def test_ring_empty_tiles_device_count_and_many_rounds(orc):
    """Tables with whole tiles that have no pair (epilogue(0) rows), a table with no pair at all, a device-side live-row count
    below the capacity (rows beyond it are not written), and a table large enough for several turns of the ring
    (cfg 5: 300 k voxels)."""
    from spx import ops
    idx_np, shape = _frame_indices(orc, 2, 2)
    dev = _dev()
    d_idx = torch.from_numpy(idx_np).to(dev)
    g = torch.Generator().manual_seed(37)
    sub = ops.subm_rulebook(d_idx, 2, shape, (3, 3, 3))
    n = sub.n_out
    holes = sub.pair.clone()
    holes[:, 128:320] = -1
    holes[:, n - 100:] = -1
    none = torch.full_like(sub.pair, -1)
    w = (torch.randn(64, 3, 3, 3, 64, generator=g) / np.sqrt(27 * 64)).to(dev)
    wp = ops.pack_weight(w, 0)
    x = torch.randn(n, 64, generator=g).to(dev)
    sh = torch.randn(64, generator=g).to(dev)
    for pair in (holes, none, sub.pair):
        plan = ops.conv_ring_plan(pair, sub.ld, 27, n)
        for relu in (False, True):
            ref = ops.conv_gemm(x, wp, 64, 27, pair, sub.ld, n, shift=sh, relu=relu)
            outs = [ops.conv_gemm_ring(x, wp, 64, 27, pair, sub.ld, n, plan, shift=sh, relu=relu) for _ in range(3)]
            assert torch.equal(outs[0], ref) and torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
        assert _plan_header(plan)["timeouts"] == 0
    out = ops.conv_gemm_ring(x, wp, 64, 27, holes, sub.ld, n, ops.conv_ring_plan(holes, sub.ld, 27, n), shift=sh, relu=True)
    assert torch.equal(out[128:320], torch.relu(sh).expand(192, 64))
    # device-side live count: the plan and the launch read it; rows beyond it keep what the buffer held
    live = n - 1234
    d_n = torch.tensor([live], dtype=torch.int64, device=dev)
    sub2 = ops.subm_rulebook(d_idx, 2, shape, (3, 3, 3), d_n=d_n)
    plan = ops.conv_ring_plan(sub2.pair, sub2.ld, 27, n, d_n)
    _check_plan(ops, plan, sub2.pair, sub2.ld, 27, n, live)
    ref = ops.conv_gemm(x, wp, 64, 27, sub2.pair, sub2.ld, n, d_n_dst=d_n)
    out = ops.conv_gemm_ring(x, wp, 64, 27, sub2.pair, sub2.ld, n, plan, d_n_dst=d_n)
    assert torch.equal(out[:live], ref[:live])
    # several rounds: 300 k rows = 18 750 tiles = 7 turns of the ring
    idx5, shape5 = _frame_indices(orc, 5, 1)
    d5 = torch.from_numpy(idx5).to(dev)
    sub5 = ops.subm_rulebook(d5, 1, shape5, (3, 3, 3))
    n5 = sub5.n_out
    x5 = torch.randn(n5, 64, generator=g).to(dev)
    plan5 = ops.conv_ring_plan(sub5.pair, sub5.ld, 27, n5)
    hdr = _check_plan(ops, plan5, sub5.pair, sub5.ld, 27, n5, n5)
    assert max(hdr["rounds"]) >= (3 if hdr["tm"] == 1 else 2)
    for flip in (False, True):
        assert torch.equal(ops.conv_gemm_ring(x5, wp, 64, 27, sub5.pair, sub5.ld, n5, plan5, flip_k=flip),
                           ops.conv_gemm(x5, wp, 64, 27, sub5.pair, sub5.ld, n5, flip_k=flip))
    assert _plan_header(plan5)["timeouts"] == 0


def test_ring_argument_checks():
    from spx import _lib
    lib = _lib.load()
    assert lib.spx_conv_ring_plan(None, 10, 27, 10, None, None, None) == -1
    assert lib.spx_conv_gemm_ring(None, 1, 64, None, 64, 27, 0, None, 10, 10, None, None, None, 0, None, None, None, None,
                                  None, None) == -1
    assert b"ring" in lib.spx_strerror(-8)
    assert lib.spx_conv_ring_stat_rows() == 256
    assert lib.spx_conv_ring_plan_bytes(100000) > 0


@pytest.mark.parametrize("ksize", [(1, 3, 3), (3, 1, 1), (1, 1, 3), (3, 3, 1), (1, 5, 5)])
def test_ring_other_kernel_volumes(ksize):
    """Kernel volumes other than the reference's 27 (3 ... 25 offsets: the offset iterator, the flipped walk of the data
    gradient and the plan's masks all depend on K), submanifold tables, bit-identical to spx_conv_gemm in both directions."""
    from spx import ops
    dev = _dev()
    K = ksize[0] * ksize[1] * ksize[2]
    g = torch.Generator().manual_seed(500 + K)
    n, side = 30000, 48
    lin = torch.randperm(side ** 3, generator=g)[:n].sort()[0]
    idx = torch.stack([torch.zeros_like(lin), lin // (side * side), (lin // side) % side, lin % side], 1).int().to(dev)
    rb = ops.subm_rulebook(idx, 1, [side, side, side], ksize)
    w = (torch.randn(64, *ksize, 64, generator=g) / np.sqrt(K * 64)).to(dev)
    wp = ops.pack_weight(w, 0)
    x = torch.randn(n, 64, generator=g).to(dev)
    plan = ops.conv_ring_plan(rb.pair, rb.ld, K, n)
    _check_plan(ops, plan, rb.pair, rb.ld, K, n, n)
    for flip in (False, True):
        ref = ops.conv_gemm(x, wp, 64, K, rb.pair, rb.ld, n, flip_k=flip)
        out = ops.conv_gemm_ring(x, wp, 64, K, rb.pair, rb.ld, n, plan, flip_k=flip)
        assert torch.equal(out, ref), (ksize, flip)
    assert _plan_header(plan)["timeouts"] == 0
    ops.check_status(dev)


@pytest.mark.parametrize("ksize,n", [((3, 1, 1), 250000), ((1, 3, 3), 250000), ((1, 1, 1), 60000)])
def test_ring_wraps_with_few_offsets(ksize, n):
    """Fewer offsets than ring slots and several turns (the ring index runs over turn * K + k: a slot is revisited within one
    turn's worth of offsets), and the degenerate single-offset table; all three tiles-per-wave settings (fixture)."""
    from spx import ops
    dev = _dev()
    K = ksize[0] * ksize[1] * ksize[2]
    g = torch.Generator().manual_seed(900 + K)
    side = 96
    lin = torch.randperm(side ** 3, generator=g)[:n].sort()[0]
    idx = torch.stack([torch.zeros_like(lin), lin // (side * side), (lin // side) % side, lin % side], 1).int().to(dev)
    rb = ops.subm_rulebook(idx, 1, [side, side, side], ksize)
    w = (torch.randn(64, *ksize, 64, generator=g) / np.sqrt(K * 64)).to(dev)
    wp = ops.pack_weight(w, 0)
    x = torch.randn(n, 64, generator=g).to(dev)
    plan = ops.conv_ring_plan(rb.pair, rb.ld, K, n)
    hdr = _check_plan(ops, plan, rb.pair, rb.ld, K, n, n)
    assert max(hdr["rounds"]) >= (2 if n > 100000 else 1)
    for flip in (False, True):
        ref = ops.conv_gemm(x, wp, 64, K, rb.pair, rb.ld, n, flip_k=flip)
        out = ops.conv_gemm_ring(x, wp, 64, K, rb.pair, rb.ld, n, plan, flip_k=flip)
        assert torch.equal(out, ref), (ksize, flip)
    assert _plan_header(plan)["timeouts"] == 0
    ops.check_status(dev)
