"""Oracle-backed CPU stand-in for `spx.ops`.  TEST INFRASTRUCTURE ONLY (see oracle/spx_oracle.c header).

`with use_oracle_backend():` temporarily replaces the functions of `spx.ops` with CPU restatements built on
oracle/oracle.py, so that the SAME module definitions (spx.modules, pcdet_amd.models) can be run end-to-end on the
host.  Two users, both allowed by the oracle rule: tests (whole-backbone / whole-detector parity of the HIP path
against the oracle, forward and backward) and bench.py's `cpu_baseline` leg.  The product never imports this file;
outside the context manager `spx.ops` refuses CPU tensors.
"""
import contextlib

import numpy as np
import torch

from . import oracle as orc


class _PackedW(object):
    def __init__(self, w, mode):
        cout, cin = w.shape[0], w.shape[-1]
        self.w = np.ascontiguousarray(w.detach().cpu().numpy().reshape(cout, -1, cin), np.float32)
        self.mode = mode


def _np(t):
    return t.detach().cpu().numpy()


def voxelize(points, point_cloud_range, voxel_size, max_points, max_voxels, batch_size=1, batch_col=-1, xyz_col=0,
             feat_col=0, num_features=None, want_voxels=True, want_mean=True, sync=True, **_kw):
    pts = np.ascontiguousarray(_np(points), np.float32)
    c = (pts.shape[1] - feat_col) if num_features is None else int(num_features)
    vs, cs, ns = [], [], []
    for b in range(batch_size):
        f = pts if batch_col < 0 else pts[pts[:, batch_col].astype(np.int64) == b]
        v, co, n = orc.voxelize(f, point_cloud_range, voxel_size, max_points, max_voxels, c=c, xyz_col=xyz_col,
                                feat_col=feat_col)
        vs.append(v)
        cs.append(np.concatenate([np.full((co.shape[0], 1), b, np.int32), co], 1))
        ns.append(n)
    v, co, n = np.concatenate(vs), np.concatenate(cs), np.concatenate(ns)
    rng = [float(x) for x in point_cloud_range]
    grid = [int(round((rng[3 + j] - rng[j]) / float(voxel_size[j]))) for j in range(3)]
    return dict(voxels=torch.from_numpy(v) if want_voxels else None, coords=torch.from_numpy(co),
                num_points=torch.from_numpy(n), mean=torch.from_numpy(orc.mean_vfe(v, n)) if want_mean else None,
                num_voxels=v.shape[0], d_num_voxels=None, grid_size=grid)


def mean_vfe(voxels, num_points):
    return torch.from_numpy(orc.mean_vfe(_np(voxels), _np(num_points).astype(np.int32)))


def _rulebook_cls():
    from spx.ops import Rulebook
    return Rulebook


def subm_rulebook(indices, batch_size, spatial_shape, ksize, dilation=(1, 1, 1), want_cnt=False, d_n=None, **_kw):
    pair, cnt = orc.subm_rulebook(_np(indices), spatial_shape, ksize, dilation)
    n = indices.shape[0]
    pair_t = torch.from_numpy(np.ascontiguousarray(pair)) if n else torch.zeros((pair.shape[0], 1), dtype=torch.int32)
    return _rulebook_cls()(pair_t, max(n, 1), n, n, pair.shape[0], True, indices, spatial_shape, spatial_shape,
                           cnt=torch.from_numpy(cnt), ksize=list(ksize), stride=[1, 1, 1],
                           padding=[k // 2 for k in ksize], dilation=list(dilation))


def conv_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=(1, 1, 1), want_cnt=False,
                  d_n_in=None, cap=None, sync=True):
    oi, pf, pb, cnt, oshape = orc.conv_rulebook(_np(indices), spatial_shape, ksize, stride, padding, dilation)
    n_in, n_out = indices.shape[0], oi.shape[0]
    return _rulebook_cls()(torch.from_numpy(np.ascontiguousarray(pf)), max(n_out, 1), n_in, n_out, pf.shape[0], False,
                           torch.from_numpy(oi), oshape, spatial_shape, pair_bwd=torch.from_numpy(pb),
                           cnt=torch.from_numpy(cnt), ksize=list(ksize), stride=list(stride), padding=list(padding),
                           dilation=list(dilation))


def pack_weight(weight, mode):
    return _PackedW(weight, mode)


def conv_gemm(src, w_packed, c_dst, kvol, pair, ld, n_dst, flip_k=False, scale=None, shift=None, relu=False,
              d_n_dst=None):
    x = _np(src).astype(np.float32)
    p = _np(pair)[:, :n_dst]
    w = w_packed.w  # [Cout, K, Cin]
    out = np.zeros((n_dst, c_dst), np.float32)
    K = w.shape[1]
    for k in range(K):
        rows = p[K - 1 - k] if flip_k else p[k]
        o = np.nonzero(rows >= 0)[0]
        if o.size == 0:
            continue
        wk = w[:, k, :]                      # [Cout, Cin]
        out[o] += x[rows[o]] @ (wk.T if w_packed.mode == 0 else wk)
    if scale is not None:
        out = out * _np(scale)
    if shift is not None:
        out = out + _np(shift)
    if relu:
        out = np.maximum(out, 0)
    return torch.from_numpy(out.astype(np.float32))


def conv_wgrad(feat_in, dout, pair, ld, n_out, wshape, d_n_out=None, counts=None):
    x, g = _np(feat_in).astype(np.float32), _np(dout).astype(np.float32)
    p = _np(pair)[:, :n_out]
    cout, cin = wshape[0], wshape[-1]
    K = p.shape[0]
    dw = np.zeros((cout, K, cin), np.float32)
    for k in range(K):
        o = np.nonzero(p[k] >= 0)[0]
        if o.size:
            dw[:, k, :] = g[o].T @ x[p[k, o]]
    return torch.from_numpy(dw.reshape(tuple(wshape)))


def densify(features, indices, batch_size, spatial_shape, channels_last=False, d_n=None):
    return torch.from_numpy(orc.densify(_np(features), _np(indices), batch_size, [int(s) for s in spatial_shape]))


def densify_bwd(ddense, indices, batch_size, spatial_shape, channels_last=False, d_n=None):
    i = indices.long()
    return ddense[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]].contiguous()


def boxes_iou_bev(boxes_a, boxes_b, overlap_only=False):
    return torch.from_numpy(orc.boxes_iou_bev(_np(boxes_a), _np(boxes_b), overlap_only))


def nms_bev(boxes_sorted, thresh, axis_aligned=False):
    keep = orc.nms_bev(_np(boxes_sorted), thresh, axis_aligned)
    out = torch.zeros((max(boxes_sorted.shape[0], 1),), dtype=torch.int64)
    out[:keep.shape[0]] = torch.from_numpy(keep)
    return out, torch.tensor([keep.shape[0]], dtype=torch.int64)


_NAMES = ["boxes_iou_bev", "nms_bev", "voxelize", "mean_vfe", "subm_rulebook", "conv_rulebook", "pack_weight", "conv_gemm", "conv_wgrad",
          "densify", "densify_bwd"]


@contextlib.contextmanager
def use_oracle_backend():
    from spx import ops
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
