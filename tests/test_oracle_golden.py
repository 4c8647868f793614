"""CPU suite: the oracle against the committed golden vectors (tests/golden/*.npz).

dense_conv_*.npz were produced by tests/golden/make_golden_dense.py from PyTorch DENSE ops only
(F.conv3d + autograd), so they pin the oracle's rulebooks (index sets, canonical order) and conv arithmetic
independently of both the oracle and the HIP library.
"""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "dense_conv_*.npz")))


def test_fixtures_present():
    assert len(FIXTURES) >= 5


@pytest.mark.parametrize("path", FIXTURES)
def test_oracle_matches_dense_conv3d(path, orc):
    d = np.load(path)
    if d["subm"]:
        pair, cnt = orc.subm_rulebook(d["idx"], d["shape"], d["k"])
        out_idx, pb = d["idx"], None
    else:
        out_idx, pair, pb, cnt, oshape = orc.conv_rulebook(d["idx"], d["shape"], d["k"], d["s"], d["p"])
        assert list(oshape) == [int(x) for x in d["out_spatial"]]
    assert np.array_equal(out_idx, d["out_idx"])          # integer, exact, in nonzero() (= canonical) order
    assert int(cnt.sum()) == int((pair >= 0).sum())
    for conv in (orc.conv_fwd, orc.conv_fwd_gemm):
        assert np.abs(conv(d["feat"], d["w"], pair) - d["out"]).max() < 2e-6
    assert np.abs(orc.conv_dgrad(d["dout"], d["w"], pair, d["idx"].shape[0]) - d["dfeat"]).max() < 5e-6
    assert np.abs(orc.conv_wgrad(d["feat"], d["dout"], pair, d["w"].shape) - d["dw"]).max() < 2e-5
    if pb is not None:  # forward and backward tables are inverse maps
        for k in range(pair.shape[0]):
            o = np.nonzero(pair[k] >= 0)[0]
            assert np.array_equal(pb[k][pair[k, o]], o)


def test_oracle_densify_bev_channel_order(orc):
    """height_compression.py:21-23: dense() then view(N, C*D, H, W)  =>  BEV channel = c*D + z."""
    rng = np.random.default_rng(0)
    idx = np.array([[0, 0, 1, 2], [0, 1, 1, 2], [1, 1, 0, 0]], np.int32)
    feat = rng.standard_normal((3, 4)).astype(np.float32)
    dense = orc.densify(feat, idx, 2, [2, 3, 4])
    bev = dense.reshape(2, 8, 3, 4)
    for r, (b, z, y, x) in enumerate(idx):
        for c in range(4):
            assert bev[b, c * 2 + z, y, x] == feat[r, c]
    assert np.count_nonzero(dense) == 12


def test_oracle_voxelize_rules(orc):
    """Sequential semantics of SURVEY.md §8a row a1 on a hand-checkable case."""
    rng_ = [0.0, 0.0, 0.0, 4.0, 4.0, 2.0]
    vs = [1.0, 1.0, 1.0]
    pts = np.array([[0.5, 0.5, 0.5, 1], [3.5, 0.5, 0.5, 2], [0.6, 0.4, 0.2, 3], [4.0, 0.0, 0.0, 4],
                    [-0.1, 0.0, 0.0, 5], [0.1, 0.9, 0.9, 6], [2.5, 2.5, 1.5, 7], [0.2, 0.2, 0.2, 8]], np.float32)
    v, c, n = orc.voxelize(pts, rng_, vs, 3, 10)
    assert c.tolist() == [[0, 0, 0], [0, 0, 3], [1, 2, 2]]      # (z,y,x), first-occurrence order
    assert n.tolist() == [3, 1, 1]                              # 4th point of voxel 0 dropped (max_points=3)
    assert v[0, :, 3].tolist() == [1, 3, 6] and v[1, 0, 3] == 2 and v[2, 0, 3] == 7
    assert np.all(v[1, 1:] == 0)
    v, c, n = orc.voxelize(pts, rng_, vs, 3, 2)                 # voxel budget 2: third voxel never created
    assert c.tolist() == [[0, 0, 0], [0, 0, 3]] and n.tolist() == [3, 1]
    m = orc.mean_vfe(v, n)
    assert np.allclose(m[0, 3], (1 + 3 + 6) / 3.0)


def test_oracle_backbone8x_shapes(orc):
    """VoxelBackBone8x wiring (spconv_backbone.py:85-125): stage shapes 41x1600x1408 -> 2x200x176."""
    from pcdet_amd.datasets import synthetic as syn
    f = syn.make_frame(1, 0)
    g = syn.KITTI
    v, c, n = orc.voxelize(f["points"], g["point_cloud_range"], g["voxel_size"], 5, 16000)
    assert v.shape[0] == 5000
    idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    rs = np.random.default_rng(0)
    weights, bn = {}, {}
    for name, cin, cout, ks, _st, _pd, _t, _key in orc.backbone8x_spec(4):
        weights[name] = (rs.standard_normal((cout, *ks, cin)) / np.sqrt(27 * cin)).astype(np.float32)
        bn[name] = (np.ones(cout, np.float32), np.zeros(cout, np.float32), np.zeros(cout), np.ones(cout))
    out = orc.backbone8x_forward(orc.mean_vfe(v, n), idx, 1, [41, 1600, 1408], weights, bn)
    assert out["conv1.0.0"][2] == [41, 1600, 1408]
    assert out["conv2.2.0"][2] == [21, 800, 704]
    assert out["conv3.2.0"][2] == [11, 400, 352]
    assert out["conv4.2.0"][2] == [5, 200, 176]
    assert out["conv_out.0"][2] == [2, 200, 176] and out["conv_out.0"][0].shape[1] == 128


def test_oracle_rotated_iou_known_answers(orc):
    """Rotated BEV IoU restatement (iou3d_cpu.cpp:66-229) on cases with closed-form answers."""
    def box(x, y, dx, dy, r):
        return [x, y, 0.0, dx, dy, 1.0, r]
    a = np.array([box(0, 0, 2, 2, 0.0)], np.float32)
    b = np.array([box(0, 0, 2, 2, 0.0), box(1, 0, 2, 2, 0.0), box(5, 5, 1, 1, 0.3), box(0, 0, 2, 2, np.pi / 4),
                  box(0, 0, 4, 1, np.pi / 2), box(0.5, 0.5, 1, 1, np.pi)], np.float32)
    iou = orc.boxes_iou_bev(a, b)[0]
    ov = orc.boxes_iou_bev(a, b, overlap_only=True)[0]
    assert abs(iou[0] - 1.0) < 1e-6
    assert abs(ov[1] - 2.0) < 1e-5 and abs(iou[1] - 2.0 / 6.0) < 1e-5          # half overlap
    assert ov[2] == 0.0 and iou[2] == 0.0                                       # disjoint
    assert abs(ov[3] - 8.0 * (np.sqrt(2.0) - 1.0)) < 1e-4                       # square vs 45-degree square: octagon
    assert abs(ov[4] - 2.0) < 1e-5                                              # 4x1 turned 90 degrees inside-crossing
    assert abs(ov[5] - 1.0) < 1e-5 and abs(iou[5] - 0.25) < 1e-5                # contained box
    # greedy NMS semantics: box 1 suppresses its heavy overlaps only
    boxes = np.array([box(0, 0, 2, 2, 0), box(0.1, 0, 2, 2, 0), box(3, 0, 2, 2, 0), box(3.05, 0.05, 2, 2, 0.1),
                      box(10, 10, 1, 1, 0)], np.float32)
    assert orc.nms_bev(boxes, 0.5).tolist() == [0, 2, 4]
    assert orc.nms_bev(boxes, 0.99).tolist() == [0, 1, 2, 3, 4]
    assert orc.nms_bev(boxes[:0], 0.5).tolist() == []


def test_dynamic_voxelize_oracle_against_torch_unique():
    """Row a5': the oracle's dynamic voxelisation against the reference's formulation written with stock torch ops on the
    CPU (torch.floor / torch.unique / index_add_ in place of torch_scatter.scatter_mean, which is not installed): keys,
    coords and inverse exactly; means to fp32 round-off."""
    import torch
    from oracle import oracle as orc
    from pcdet_amd.datasets import synthetic
    b = synthetic.make_batch(0, 3)
    pts = torch.from_numpy(b["points"])
    geom = synthetic.CONFIGS[0]["geom"]
    rng, vs = geom["point_cloud_range"], geom["voxel_size"]
    grid = torch.tensor([int(round((rng[3 + j] - rng[j]) / vs[j])) for j in range(3)])
    pc = torch.floor((pts[:, 1:4] - torch.tensor(rng[:3])) / torch.tensor(vs)).int()          # dynamic_mean_vfe.py:51
    mask = ((pc >= 0) & (pc < grid)).all(dim=1)                                                # :52
    p2, pc2 = pts[mask], pc[mask].long()
    merge = p2[:, 0].long() * int(grid.prod()) + pc2[:, 0] * int(grid[1] * grid[2]) + pc2[:, 1] * int(grid[2]) + pc2[:, 2]
    unq, inv, cnt = torch.unique(merge, return_inverse=True, return_counts=True)               # :59
    mean = torch.zeros(unq.shape[0], 4).index_add_(0, inv, p2[:, 1:]) / cnt[:, None]           # scatter_mean, :61
    coords = torch.stack((unq // int(grid.prod()), (unq % int(grid.prod())) // int(grid[1] * grid[2]),
                          (unq % int(grid[1] * grid[2])) // int(grid[2]), unq % int(grid[2])), 1)[:, [0, 3, 2, 1]]  # :64-68
    f, c, inverse = orc.dynamic_voxelize(b["points"], rng, vs, batch_size=3)
    assert np.array_equal(c, coords.int().numpy())
    assert np.array_equal(inverse[mask.numpy()], inv.int().numpy()) and (inverse[~mask.numpy()] == -1).all()
    assert np.abs(f - mean.numpy()).max() < 1e-5


@pytest.mark.parametrize("strides,former", [((1, 1, 1), 0.0), ((1, 2, 2), 0.15), ((2, 3, 1), 0.0)])
def test_voxel_query_oracle_against_bruteforce(strides, former, orc):
    """The C restatement of voxel_query(_dilated)_kernel_stack (reference voxel_query_gpu.cu:10-100, 125-215) against a
    direct numpy scan, with nsample larger than any ball so that no random number is drawn: neighbour lists in scan
    order, first-neighbour padding, empty marker, occupied-cell and filled-slot counts."""
    rng = np.random.default_rng(3)
    B, Z, Y, X = 2, 5, 12, 12
    occ = rng.random((B, Z, Y, X)) < 0.3
    coords = np.argwhere(occ).astype(np.int32)                               # batch-major (b, z, y, x)
    n = coords.shape[0]
    table = np.full((B, Z, Y, X), -1, np.int32)
    table[tuple(coords.T)] = np.arange(n, dtype=np.int32)
    xyz = (coords[:, [3, 2, 1]].astype(np.float32) + 0.5) * np.float32(0.1)
    q = rng.choice(n, 60, replace=False)
    q_coords, q_xyz = coords[q], xyz[q] + np.float32(0.01)
    ranges, radius, nsample = (2, 3, 3), np.float32(0.32), 400
    idx, cnt, filled = orc.voxel_query_dilated(q_xyz, xyz, q_coords, table, nsample, former, radius, ranges, strides)
    for i in range(len(q)):
        b, z0, y0, x0 = q_coords[i]
        found, scanned = [], 0
        for dz in range(-ranges[0], ranges[0] + 1, strides[0]):
            for dy in range(-ranges[1], ranges[1] + 1, strides[1]):
                for dx in range(-ranges[2], ranges[2] + 1, strides[2]):
                    z, y, x = z0 + dz, y0 + dy, x0 + dx
                    if not (0 <= z < Z and 0 <= y < Y and 0 <= x < X) or table[b, z, y, x] < 0:
                        continue
                    scanned += 1
                    d = xyz[table[b, z, y, x]] - q_xyz[i]
                    d2 = np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
                    if d2 > radius * radius or d2 < np.float32(former) * np.float32(former):
                        continue
                    found.append(int(table[b, z, y, x]))
        assert cnt[i] == scanned and filled[i] == len(found) < nsample
        if found:
            want = (found + found * (nsample // len(found) + 1))[:nsample]           # cyclic padding == idx[cnt] = idx[l]
            assert idx[i].tolist() == want
        else:
            assert idx[i, 0] == -1 and (idx[i, 1:] == 0).all()             # the padding loop copies slot l onto itself
    if strides == (1, 1, 1) and former == 0.0:
        i_plain, c_plain = orc.voxel_query(q_xyz, xyz, q_coords, table, nsample, radius, ranges)
        assert np.array_equal(i_plain, idx) and np.array_equal(c_plain, cnt)


@pytest.mark.parametrize("stride", [1, 2])
def test_bev_entry_identity_on_cpu(stride, orc):
    """The identity BaseBEVBackbone's sparse entry relies on (pcdet_amd/models/backbones_2d/base_bev_backbone.py:
    _sparse_entry), checked with the CPU oracle and torch-CPU only: ZeroPad2d(1) + Conv2d(C*D, Co, 3, stride s, bias=False)
    over HeightCompression's map [B, C*D, H, W] (channel = c*D + z, reference height_compression.py:21-23) equals the sparse
    convolution of the encoded tensor with kernel (D,3,3), stride (1,s,s), padding (0,1,1) and weight
    W3[co, z, ky, kx, c] = W2[co, c*D + z, ky, kx], scattered to the output pixels — and is exactly zero elsewhere."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(11 + stride)
    B, D, H, W, C, CO = 2, 2, 14, 12, 6, 5
    occ = rng.random((B, D, H, W)) < 0.15
    idx = np.argwhere(occ).astype(np.int32)                                  # (b, z, y, x) ascending
    feats = rng.standard_normal((idx.shape[0], C)).astype(np.float32)
    w2 = rng.standard_normal((CO, C * D, 3, 3)).astype(np.float32)
    dense = np.zeros((B, C, D, H, W), np.float32)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feats
    bev = torch.from_numpy(dense.reshape(B, C * D, H, W))
    yd = F.conv2d(bev, torch.from_numpy(w2), None, stride, 1).numpy()       # [B, CO, Ho, Wo]
    out_idx, pair_f, _pb, _cnt, out_shape = orc.conv_rulebook(idx, [D, H, W], (D, 3, 3), (1, stride, stride), (0, 1, 1))
    assert list(out_shape) == [1, yd.shape[2], yd.shape[3]] and (out_idx[:, 1] == 0).all()
    w3 = np.ascontiguousarray(w2.reshape(CO, C, D, 3, 3).transpose(0, 2, 3, 4, 1))        # [co, z, ky, kx, c]
    ys = orc.conv_fwd(feats, w3.reshape(CO, D * 9, C), pair_f)
    got = yd.transpose(0, 2, 3, 1)[out_idx[:, 0], out_idx[:, 2], out_idx[:, 3]]
    assert np.abs(ys - got).max() < 1e-5 * max(1.0, np.abs(got).max())
    covered = np.zeros((B, yd.shape[2], yd.shape[3]), bool)
    covered[out_idx[:, 0], out_idx[:, 2], out_idx[:, 3]] = True
    assert np.abs(yd.transpose(0, 2, 3, 1)[~covered]).max() == 0.0
