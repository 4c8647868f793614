"""CPU ORACLE (python face).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product package never does.  See the header of ``oracle/spx_oracle.c`` for what is
restated, what pins it, and why parity against real spconv row order is "parity unpinned".

Two layers:
  * thin ctypes wrappers over ``oracle/spx_oracle.c`` (integer-exact rulebooks, voxeliser, loop convs);
  * numpy restatements of the per-offset gather -> GEMM -> scatter-add algorithm (what spconv's CPU
    "Native" path does) and of the VoxelBackBone8x wiring
    (reference pcdet/models/backbones_3d/spconv_backbone.py:77-194) used for whole-backbone parity
    and as the multi-threaded CPU baseline.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "spx_oracle.c")
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "libspx_oracle.so")
_lib = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """gcc-compile the C restatement into oracle/_build/ (building the checker is not using it)."""
    os.makedirs(_BUILD, exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_voxelize.restype = ctypes.c_int64
        _lib.orc_conv_rulebook.restype = ctypes.c_int64
    return _lib


def _f(a):
    return a.ctypes.data_as(c_f32p)


def _i(a):
    return a.ctypes.data_as(c_i32p)


def _i3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.int32).reshape(3))


def out_shape_of(in_shape, ksize, stride, pad, dil):
    """floor((in + 2p - d(k-1) - 1)/s) + 1 per axis (the F.conv3d / spconv rule)."""
    return [(int(i) + 2 * int(p) - int(d) * (int(k) - 1) - 1) // int(s) + 1
            for i, k, s, p, d in zip(in_shape, ksize, stride, pad, dil)]


# --------------------------------------------------------------------------- C wrappers

def voxelize(points, rng, vsize, max_points, max_voxels, c=None, xyz_col=0, feat_col=0):
    """One frame.  Returns voxels[M,T,C], coords[M,3](z,y,x), num[M].  data_processor.py:127-155."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, stride = points.shape
    c = stride - feat_col if c is None else c
    rng = np.asarray(rng, dtype=np.float32)
    vsize = np.asarray(vsize, dtype=np.float32)
    grid = np.round((rng[3:6].astype(np.float64) - rng[0:3].astype(np.float64)) / vsize.astype(np.float64)).astype(np.int32)
    voxels = np.zeros((max_voxels, max_points, c), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    m = lib().orc_voxelize(_f(points), ctypes.c_int64(n), stride, xyz_col, feat_col, c, _f(rng), _f(vsize),
                           _i(grid), max_points, max_voxels, _f(voxels), _i(coords), _i(num))
    return voxels[:m].copy(), coords[:m].copy(), num[:m].copy()


def dynamic_voxelize(points, rng, vsize, batch_size=1, batch_col=0, xyz_col=1, num_features=None):
    """DynamicMeanVFE.forward restated (reference pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:48-72): cells by
    floor((xyz - range_min) / voxel_size) in fp32, in-range mask, merge key ((b*X + cx)*Y + cy)*Z + cz, unique (sorted),
    per-voxel mean of the columns [xyz_col, xyz_col + C) — summed sequentially in point order in fp32 (the reference
    sums with float atomics, so its own bits are not defined).  Returns features [M, C] f32,
    coords [M, 4] int32 (b, z, y, x), inverse [N] int32 (-1 = dropped)."""
    pts = np.ascontiguousarray(points, np.float32)
    n, stride = pts.shape
    c = (stride - xyz_col) if num_features is None else int(num_features)
    lo = np.asarray(rng[:3], np.float32)
    vs = np.asarray(vsize, np.float32)
    grid = np.array([int(round((float(rng[3 + j]) - float(rng[j])) / float(vsize[j]))) for j in range(3)], np.int64)
    f = np.floor((pts[:, xyz_col:xyz_col + 3] - lo) / vs)
    b = pts[:, batch_col].astype(np.int64) if batch_col >= 0 else np.zeros(n, np.int64)
    ok = np.all((f >= 0) & (f < grid.astype(np.float32)), axis=1) & (b >= 0) & (b < batch_size)
    cell = f.astype(np.int64)
    key = ((b * grid[0] + cell[:, 0]) * grid[1] + cell[:, 1]) * grid[2] + cell[:, 2]
    uniq, inv_ok = np.unique(key[ok], return_inverse=True)
    inverse = np.full(n, -1, np.int32)
    inverse[ok] = inv_ok.astype(np.int32)
    m = uniq.shape[0]
    sums = np.zeros((m, c), np.float32)
    np.add.at(sums, inv_ok, pts[ok][:, xyz_col:xyz_col + c])      # unbuffered: sequential in point order
    counts = np.bincount(inv_ok, minlength=m).astype(np.float32)
    feats = sums * (np.float32(1.0) / counts)[:, None]
    z = uniq % grid[2]
    y = (uniq // grid[2]) % grid[1]
    x = (uniq // (grid[2] * grid[1])) % grid[0]
    bb = uniq // (grid[2] * grid[1] * grid[0])
    coords = np.stack([bb, z, y, x], 1).astype(np.int32)
    return feats.astype(np.float32), coords, inverse


def mean_vfe(voxels, num):
    voxels = np.ascontiguousarray(voxels, np.float32)
    num = np.ascontiguousarray(num, np.int32)
    n, t, c = voxels.shape
    out = np.zeros((n, c), np.float32)
    lib().orc_mean_vfe(_f(voxels), _i(num), ctypes.c_int64(n), t, c, _f(out))
    return out


def subm_rulebook(idx, shape, ksize=(3, 3, 3), dil=(1, 1, 1)):
    idx = np.ascontiguousarray(idx, np.int32)
    n = idx.shape[0]
    K = int(np.prod(ksize))
    pair = np.full((K, max(n, 1)), -1, np.int32)
    cnt = np.zeros((K,), np.int32)
    lib().orc_subm_rulebook(_i(idx), ctypes.c_int64(n), _i(_i3(shape)), _i(_i3(ksize)), _i(_i3(dil)), _i(pair),
                            ctypes.c_int64(max(n, 1)), _i(cnt))
    return pair[:, :n].copy(), cnt


def conv_rulebook(idx, in_shape, ksize, stride, pad, dil=(1, 1, 1)):
    """Returns out_idx[n_out,4], pair_fwd[K,n_out], pair_bwd[K,n_in], cnt[K], out_shape."""
    idx = np.ascontiguousarray(idx, np.int32)
    n_in = idx.shape[0]
    K = int(np.prod(ksize))
    out_shape = out_shape_of(in_shape, ksize, stride, pad, dil)
    cap = max(int(n_in) * K, 1)
    out_idx = np.zeros((cap, 4), np.int32)
    pf = np.full((K, cap), -1, np.int32)
    pb = np.full((K, max(n_in, 1)), -1, np.int32)
    cnt = np.zeros((K,), np.int32)
    n_out = lib().orc_conv_rulebook(_i(idx), ctypes.c_int64(n_in), _i(_i3(in_shape)), _i(_i3(out_shape)),
                                    _i(_i3(ksize)), _i(_i3(stride)), _i(_i3(pad)), _i(_i3(dil)), _i(out_idx),
                                    _i(pf), _i(pb), _i(cnt), ctypes.c_int64(cap))
    assert n_out >= 0
    return out_idx[:n_out].copy(), pf[:, :n_out].copy(), pb[:, :n_in].copy(), cnt, out_shape


def conv_fwd(feat, w, pair, acc64=False):
    """feat[N_in,Cin], w[Cout,K,Cin] (or [Cout,kz,ky,kx,Cin]), pair[K,N_out] -> out[N_out,Cout]."""
    feat = np.ascontiguousarray(feat, np.float32)
    cout, cin = w.shape[0], w.shape[-1]
    w = np.ascontiguousarray(w, np.float32).reshape(cout, -1, cin)
    K = w.shape[1]
    pair = np.ascontiguousarray(pair, np.int32)
    n_out = pair.shape[1]
    out = np.zeros((n_out, cout), np.float32)
    lib().orc_conv_fwd(_f(feat), cin, _f(w), cout, K, _i(pair), ctypes.c_int64(max(n_out, 1) if n_out == 0 else n_out),
                       ctypes.c_int64(n_out), _f(out), int(bool(acc64)))
    return out


def conv_dgrad(dout, w, pair, n_in):
    dout = np.ascontiguousarray(dout, np.float32)
    cout, cin = w.shape[0], w.shape[-1]
    w = np.ascontiguousarray(w, np.float32).reshape(cout, -1, cin)
    K = w.shape[1]
    pair = np.ascontiguousarray(pair, np.int32)
    n_out = pair.shape[1]
    din = np.zeros((n_in, cin), np.float32)
    lib().orc_conv_dgrad(_f(dout), cout, _f(w), cin, K, _i(pair), ctypes.c_int64(max(n_out, 1)), ctypes.c_int64(n_out),
                         ctypes.c_int64(n_in), _f(din))
    return din


def conv_wgrad(feat, dout, pair, wshape):
    feat = np.ascontiguousarray(feat, np.float32)
    dout = np.ascontiguousarray(dout, np.float32)
    cout, cin = wshape[0], wshape[-1]
    K = int(np.prod(wshape[1:-1]))
    pair = np.ascontiguousarray(pair, np.int32)
    n_out = pair.shape[1]
    dw = np.zeros((cout, K, cin), np.float32)
    lib().orc_conv_wgrad(_f(feat), cin, _f(dout), cout, K, _i(pair), ctypes.c_int64(max(n_out, 1)),
                         ctypes.c_int64(n_out), _f(dw))
    return dw.reshape(wshape)


def densify(feat, idx, batch, shape):
    feat = np.ascontiguousarray(feat, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    n, c = feat.shape
    dense = np.zeros((batch, c, shape[0], shape[1], shape[2]), np.float32)
    lib().orc_densify(_f(feat), _i(idx), ctypes.c_int64(n), c, batch, _i(_i3(shape)), _f(dense))
    return dense


# --------------------------------------------------------------------------- numpy restatements

def conv_fwd_gemm(feat, w, pair):
    """Per-offset gather -> GEMM -> scatter-add (spconv CPU 'Native' algorithm)."""
    cout, cin = w.shape[0], w.shape[-1]
    w = np.asarray(w, np.float32).reshape(cout, -1, cin)
    n_out = pair.shape[1]
    out = np.zeros((n_out, cout), np.float32)
    for k in range(w.shape[1]):
        o = np.nonzero(pair[k] >= 0)[0]
        if o.size == 0:
            continue
        out[o] += feat[pair[k, o]] @ w[:, k, :].T
    return out


def batchnorm1d(x, gamma, beta, mean=None, var=None, eps=1e-3):
    """nn.BatchNorm1d(eps=1e-3) as built at spconv_backbone.py:81; batch stats when mean is None."""
    if mean is None:
        mean = x.mean(0, dtype=np.float64)
        var = x.var(0, dtype=np.float64)
    y = (x.astype(np.float64) - mean) / np.sqrt(var + eps) * gamma + beta
    return y.astype(np.float32)


#: (name, cin_key, cout, ksize, stride, pad, conv_type, indice_key) — spconv_backbone.py:85-125
def backbone8x_spec(input_channels, last_pad=0):
    return [
        ("conv_input.0", input_channels, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm1"),
        ("conv1.0.0", 16, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm1"),
        ("conv2.0.0", 16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), "spconv", "spconv2"),
        ("conv2.1.0", 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm2"),
        ("conv2.2.0", 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm2"),
        ("conv3.0.0", 32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), "spconv", "spconv3"),
        ("conv3.1.0", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm3"),
        ("conv3.2.0", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm3"),
        ("conv4.0.0", 64, 64, (3, 3, 3), (2, 2, 2), (0, 1, 1), "spconv", "spconv4"),
        ("conv4.1.0", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm4"),
        ("conv4.2.0", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), "subm", "subm4"),
        ("conv_out.0", 64, 128, (3, 1, 1), (2, 1, 1), (last_pad,) * 3, "spconv", "spconv_down2"),
    ]


def backbone8x_forward(feat, idx, batch, sparse_shape, weights, bn, train_bn=False, conv=conv_fwd_gemm,
                       last_pad=0):
    """VoxelBackBone8x.forward (spconv_backbone.py:138-194) on the CPU.

    weights: {conv name: w[Cout,kz,ky,kx,Cin]}; bn: {conv name: (gamma, beta, running_mean, running_var)}.
    Returns dict layer-name -> (features, indices, spatial_shape) for every conv, plus 'rulebooks'.
    """
    out = {}
    books = {}
    shape = list(sparse_shape)
    cur_f, cur_i = np.asarray(feat, np.float32), np.asarray(idx, np.int32)
    for name, _cin, _cout, ks, st, pd, ctype, key in backbone8x_spec(feat.shape[1], last_pad):
        if ctype == "subm":
            if key not in books:
                books[key] = subm_rulebook(cur_i, shape, ks)[0]
            pair = books[key]
        else:
            o_idx, pair, _pb, _cnt, shape = conv_rulebook(cur_i, shape, ks, st, pd)
            books[key] = pair
            cur_i = o_idx
        y = conv(cur_f, weights[name], pair)
        g, b, rm, rv = bn[name]
        y = batchnorm1d(y, g, b, None if train_bn else rm, None if train_bn else rv)
        cur_f = np.maximum(y, 0.0)
        out[name] = (cur_f, cur_i, list(shape))
    out["rulebooks"] = books
    return out


# --------------------------------------------------------------------------- f-1: rotated BEV IoU / NMS

def boxes_iou_bev(a, b, overlap_only=False):
    a = np.ascontiguousarray(a, np.float32)[:, :7].copy()
    b = np.ascontiguousarray(b, np.float32)[:, :7].copy()
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_iou_bev(_f(a), ctypes.c_int64(a.shape[0]), _f(b), ctypes.c_int64(b.shape[0]), int(overlap_only), _f(out))
    return out


def nms_bev(boxes_sorted, thresh, axis_aligned=False):
    """boxes already sorted by descending score -> kept positions (ascending)."""
    b = np.ascontiguousarray(boxes_sorted, np.float32)[:, :7].copy()
    keep = np.zeros((max(b.shape[0], 1),), np.int64)
    lib().orc_nms_bev.restype = ctypes.c_int64
    n = lib().orc_nms_bev(_f(b), ctypes.c_int64(b.shape[0]), ctypes.c_float(thresh), int(axis_aligned),
                          keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:n].copy()


def voxel_query(new_xyz, xyz, new_coords, point_indices, nsample, radius, ranges):
    """voxel_query_kernel_stack restated (reference voxel_query_gpu.cu:10-100; oracle/spx_oracle.c: the reservoir step's
    generator is PARITY UNPINNED).  Returns idx [M, nsample] int32 (raw: -1 in slot 0 for empty balls) and the number of
    occupied cells scanned per query."""
    q = np.ascontiguousarray(new_xyz, np.float32)
    p = np.ascontiguousarray(xyz, np.float32)
    c = np.ascontiguousarray(new_coords, np.int32)
    t = np.ascontiguousarray(point_indices, np.int32)
    m = q.shape[0]
    idx = np.zeros((max(m, 1), nsample), np.int32)
    cnt = np.zeros((max(m, 1),), np.int32)
    _b, r1, r2, r3 = t.shape
    lib().orc_voxel_query(_f(q), _f(p), _i(c), _i(t), ctypes.c_int64(m), int(r1), int(r2), int(r3), int(nsample),
                          ctypes.c_float(radius), int(ranges[0]), int(ranges[1]), int(ranges[2]), _i(idx), _i(cnt))
    return idx[:m], cnt[:m]


def voxel_query_dilated(new_xyz, xyz, new_coords, point_indices, nsample, former_radius, radius, ranges, strides):
    """voxel_query_dilated_kernel_stack restated (reference voxel_query_gpu.cu:125-215; same PARITY UNPINNED generator).
    Returns idx [M, nsample], cnt_unique [M], idx_cnt [M]."""
    q = np.ascontiguousarray(new_xyz, np.float32)
    p = np.ascontiguousarray(xyz, np.float32)
    c = np.ascontiguousarray(new_coords, np.int32)
    t = np.ascontiguousarray(point_indices, np.int32)
    m = q.shape[0]
    idx = np.zeros((max(m, 1), nsample), np.int32)
    cnt = np.zeros((max(m, 1),), np.int32)
    filled = np.zeros((max(m, 1),), np.int32)
    _b, r1, r2, r3 = t.shape
    lib().orc_voxel_query_dilated(_f(q), _f(p), _i(c), _i(t), ctypes.c_int64(m), int(r1), int(r2), int(r3), int(nsample),
                                  ctypes.c_float(former_radius), ctypes.c_float(radius), int(ranges[0]), int(ranges[1]),
                                  int(ranges[2]), int(strides[0]), int(strides[1]), int(strides[2]), _i(idx), _i(cnt),
                                  _i(filled))
    return idx[:m], cnt[:m], filled[:m]
