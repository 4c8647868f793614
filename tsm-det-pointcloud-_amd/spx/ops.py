"""Tensor-level wrappers over the C ABI (include/spx.h).  torch is used for device memory and streams only.

Every function here launches HIP kernels from libspx.so on torch's current stream; nothing is computed by
torch ops and there is no CPU path (inputs must live on a GPU).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, f_arr, i3

_WS = {}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the stream handle without building a Stream object


def _stream_handle(device):
    """torch's current stream on `device` as an integer handle (several microseconds cheaper per call than
    torch.cuda.current_stream(): the small layers at the end of the backward pass are host bound)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def _stream(t):
    return ctypes.c_void_p(_stream_handle(t.device))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SpxError("spx ops need GPU tensors (got a %s tensor): the sparse-conv hot path has no "
                                "CPU fallback" % t.device)


def workspace(device, nbytes):
    """Grow-only scratch buffer per (device, stream); stream order makes back-to-back reuse safe."""
    key = (device.index, _stream_handle(device))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


_STATUS = {}


def status_word(device):
    """The sticky device status word of `device` (int32[1], include/spx.h: d_status): kernels of calls made without a host
    read (static-capacity / graph mode) report device-side errors here; check_status() reads it."""
    w = _STATUS.get(device.index)
    if w is None:
        w = _STATUS[device.index] = torch.zeros((1,), dtype=torch.int32, device=device)
    return w


def check_status(device):
    """Read (one host sync) and reset the sticky status word; raises SpxError on a device-side error."""
    w = status_word(device)
    rc = int(w.item())
    if rc != 0:
        w.zero_()
        check(rc, "device status")


def _count_and_status(buf, what):
    """buf: int64[2] = (row count, status word in the low 32 bits) -> count; raises on a device-side error (one sync)."""
    cnt, st = buf.tolist()
    st &= 0xFFFFFFFF
    if st:
        check(st - (1 << 32), what)
    return int(cnt)


def out_shape_of(in_shape, ksize, stride, pad, dil):
    return [(int(i) + 2 * int(p) - int(d) * (int(k) - 1) - 1) // int(s) + 1
            for i, k, s, p, d in zip(in_shape, ksize, stride, pad, dil)]


class Rulebook(object):
    """Indice pairs of one sparse conv (what spconv caches under `indice_key`).

    pair      int32 [K, ld]   forward table: pair[k, o] = input row feeding output o at offset k (or -1)
    pair_bwd  int32 [K, n_in] backward table (None for submanifold: the forward table is read at K-1-k)
    """

    def __init__(self, pair, ld, n_in, n_out, kvol, subm, out_indices, out_shape, in_shape, pair_bwd=None, cnt=None,
                 ksize=None, stride=None, padding=None, dilation=None):
        self.pair, self.ld, self.n_in, self.n_out, self.kvol, self.subm = pair, ld, n_in, n_out, kvol, subm
        self.out_indices, self.out_shape, self.in_shape = out_indices, list(out_shape), list(in_shape)
        self._plans = {}
        self.pair_bwd, self.cnt = pair_bwd, cnt
        self.ksize, self.stride, self.padding, self.dilation = ksize, stride, padding, dilation
        # static-capacity (graph) mode: live row counts stay on the device, n_in / n_out above are CAPACITIES
        self.d_n_in = self.d_n_out = None


# ------------------------------------------------------------------------------------------- voxelise

def voxelize(points, point_cloud_range, voxel_size, max_points, max_voxels, batch_size=1, batch_col=-1, xyz_col=0,
             feat_col=0, num_features=None, want_voxels=True, want_mean=True, sync=True, ws_precleared=False):
    """GPU hard voxelisation (+ fused MeanVFE).  Returns dict(voxels, coords[M,4], num_points, mean, M).

    sync=True: one host sync (reads M) and exact-size views.  sync=False (static-capacity / graph mode): no sync, every
    output keeps its capacity rows and the live count is the device tensor `d_num_voxels`.
    ws_precleared: SPX_WS_PRECLEARED — the caller filled ops.workspace() itself (see include/spx.h).
    Semantics: include/spx.h §1 / SURVEY.md §8a row a1.
    """
    _need_gpu(points)
    lib = _lib.load()
    points = points.contiguous().float()
    n, stride = points.shape
    c = (stride - feat_col) if num_features is None else int(num_features)
    rng = [float(x) for x in point_cloud_range]
    vs = [float(x) for x in voxel_size]
    grid = [int(round((rng[3 + j] - rng[j]) / vs[j])) for j in range(3)]
    cap = max(1, min(n, batch_size * max_voxels))
    dev = points.device
    voxels = torch.empty((cap, max_points, c), dtype=torch.float32, device=dev) if want_voxels else None
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, c), dtype=torch.float32, device=dev) if want_mean else None
    # (count, status): with a host read the call gets its own status word next to the count, one copy fetches both;
    # without, device-side errors go to the sticky per-device word (check_status)
    buf = torch.zeros((2,), dtype=torch.int64, device=dev)
    d_m = buf[:1]
    d_st = ctypes.c_void_p(buf.data_ptr() + 8) if sync else _ptr(status_word(dev))
    wsb = lib.spx_voxelize_ws_bytes(n, batch_size, max_points)
    ws = workspace(dev, wsb)
    check(lib.spx_voxelize(_ptr(points), n, stride, xyz_col, feat_col, c, batch_col, batch_size, f_arr(rng), f_arr(vs),
                           i3(grid), max_points, max_voxels, _ptr(voxels), _ptr(coords), _ptr(num), _ptr(mean),
                           _ptr(d_m), cap, 1 if ws_precleared else 0, d_st, _ptr(ws), wsb, _stream(points)),
          "spx_voxelize")
    if not sync:
        return dict(voxels=voxels, coords=coords, num_points=num, mean=mean, num_voxels=None, d_num_voxels=d_m,
                    grid_size=grid)
    m = _count_and_status(buf, "spx_voxelize")
    return dict(voxels=None if voxels is None else voxels[:m], coords=coords[:m], num_points=num[:m],
                mean=None if mean is None else mean[:m], num_voxels=m, d_num_voxels=d_m, grid_size=grid)


def dynamic_voxelize(points, point_cloud_range, voxel_size, batch_size=1, batch_col=0, xyz_col=1, num_features=None,
                     max_voxels=None, sync=True):
    """Dynamic voxelisation + per-voxel mean (include/spx.h §1b; reference DynamicMeanVFE.forward).  Returns
    dict(features [M, C], coords [M, 4] (b, z, y, x), inverse [N] (voxel row of each point, -1 = out of range),
    num_voxels, d_num_voxels).  Voxels are sorted by the key ((b*X + cx)*Y + cy)*Z + cz."""
    _need_gpu(points)
    lib = _lib.load()
    points = points.contiguous().float()
    n, stride = points.shape
    c = (stride - xyz_col) if num_features is None else int(num_features)
    rng = [float(x) for x in point_cloud_range]
    vs = [float(x) for x in voxel_size]
    grid = [int(round((rng[3 + j] - rng[j]) / vs[j])) for j in range(3)]
    cap = max(1, n if max_voxels is None else min(n, int(max_voxels)))
    dev = points.device
    feats = torch.empty((cap, c), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    inv = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    d_num = torch.zeros((1,), dtype=torch.int64, device=dev)
    g3 = i3(grid)
    wsb = lib.spx_dynamic_voxelize_ws_bytes(n, batch_size, g3, cap)
    ws = workspace(dev, wsb)
    check(lib.spx_dynamic_voxelize(_ptr(points), n, stride, batch_col, xyz_col, c, f_arr(rng), f_arr(vs), g3, batch_size,
                                   _ptr(feats), _ptr(coords), _ptr(inv), _ptr(d_num), cap, _ptr(ws), wsb, _stream(points)),
          "spx_dynamic_voxelize")
    out = {"d_num_voxels": d_num, "grid_size": grid}
    if sync:
        m = min(int(d_num.item()), cap)
        out.update(features=feats[:m], coords=coords[:m], inverse=inv[:n], num_voxels=m)
    else:
        out.update(features=feats, coords=coords, inverse=inv[:n], num_voxels=None)
    return out


def mean_vfe(voxels, num_points):
    _need_gpu(voxels, num_points)
    lib = _lib.load()
    voxels = voxels.contiguous().float()
    num = num_points.contiguous().to(torch.int32)
    n, t, c = voxels.shape
    out = torch.empty((n, c), dtype=torch.float32, device=voxels.device)
    check(lib.spx_mean_vfe(_ptr(voxels), _ptr(num), n, None, t, c, _ptr(out), _stream(voxels)), "spx_mean_vfe")
    return out


# ------------------------------------------------------------------------------------------- rulebooks

def subm_rulebook(indices, batch_size, spatial_shape, ksize, dilation=(1, 1, 1), want_cnt=False, d_n=None,
                  ws_precleared=False, unique=False):
    """d_n: optional device int64[1] live row count (<= indices.shape[0], which is then the capacity).  Device-side
    errors (only possible with ws_precleared on a dirty workspace) go to the sticky status word: check_status().
    unique: the caller guarantees one row per cell (SPX_ROWS_UNIQUE: half the table is probed, hits written twice)."""
    _need_gpu(indices)
    lib = _lib.load()
    indices = indices.contiguous()
    assert indices.dtype == torch.int32 and indices.shape[1] == 4
    n = indices.shape[0]
    K = int(ksize[0]) * int(ksize[1]) * int(ksize[2])
    dev = indices.device
    ld = max(n, 1)
    pair = torch.empty((K, ld), dtype=torch.int32, device=dev)
    cnt = torch.empty((K,), dtype=torch.int32, device=dev) if want_cnt else None   # zeroed by the library
    wsb = lib.spx_subm_rulebook_ws_bytes(n)
    ws = workspace(dev, wsb)
    check(lib.spx_subm_rulebook(_ptr(indices), n, _ptr(d_n), batch_size, i3(spatial_shape), i3(ksize), i3(dilation),
                                _ptr(pair), ld, _ptr(cnt), (1 if ws_precleared else 0) | (2 if unique else 0), _ptr(status_word(dev)),
                                _ptr(ws), wsb,
                                _stream(indices)), "spx_subm_rulebook")
    rb = Rulebook(pair, ld, n, n, K, True, indices, spatial_shape, spatial_shape, cnt=cnt, ksize=list(ksize),
                  stride=[1, 1, 1], padding=[k // 2 for k in ksize], dilation=list(dilation))
    rb.d_n_in = rb.d_n_out = d_n
    return rb


def conv_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=(1, 1, 1), want_cnt=False,
                  d_n_in=None, cap=None, sync=True, subm_ksize=None, subm_dilation=(1, 1, 1)):
    """Regular sparse conv rulebook.  subm_ksize: also build the SUBMANIFOLD table of the output level from the same rank
    bitmap (no hash for that level); it comes back as `rb.subm_next` (a Rulebook over rb.out_indices).  sync=True: one host sync (reads n_out), exact-size out_indices; sync="later": a
    PendingRulebook whose .finish() does that read.
    sync=False (static-capacity / graph mode): no sync; `cap` output rows (default: the no-overflow bound), live counts
    in rb.d_n_in / rb.d_n_out; rows beyond cap are dropped (overflow <=> rb.d_n_out > cap)."""
    _need_gpu(indices)
    lib = _lib.load()
    indices = indices.contiguous()
    assert indices.dtype == torch.int32 and indices.shape[1] == 4
    n_in = indices.shape[0]
    K = int(ksize[0]) * int(ksize[1]) * int(ksize[2])
    dev = indices.device
    out_shape = out_shape_of(spatial_shape, ksize, stride, padding, dilation)
    safe_cap = int(lib.spx_conv_out_cap(n_in, batch_size, i3(out_shape), i3(ksize), i3(stride)))
    cap = safe_cap if cap is None else max(1, min(int(cap), safe_cap))
    out_idx = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    pair_fwd = torch.empty((K, cap), dtype=torch.int32, device=dev)
    pair_bwd = torch.empty((K, max(n_in, 1)), dtype=torch.int32, device=dev)
    cnt = torch.empty((K,), dtype=torch.int32, device=dev) if want_cnt else None   # zeroed by the library
    d_n = torch.zeros((1,), dtype=torch.int64, device=dev)
    Ks = 0 if subm_ksize is None else int(subm_ksize[0]) * int(subm_ksize[1]) * int(subm_ksize[2])
    subm_pair = torch.empty((Ks, cap), dtype=torch.int32, device=dev) if Ks else None
    subm_cnt = torch.empty((Ks,), dtype=torch.int32, device=dev) if (Ks and want_cnt) else None
    wsb = lib.spx_conv_rulebook_ws_bytes(n_in, batch_size, i3(out_shape))
    ws = workspace(dev, wsb)
    # a capacity below the no-overflow bound can drop rows: the kernels say so in the sticky status word (check_status)
    d_st = _ptr(status_word(dev)) if cap < safe_cap else None
    check(lib.spx_conv_rulebook(_ptr(indices), n_in, _ptr(d_n_in), batch_size, i3(spatial_shape), i3(out_shape), i3(ksize),
                                i3(stride), i3(padding), i3(dilation), _ptr(out_idx), _ptr(pair_fwd), _ptr(pair_bwd),
                                _ptr(cnt), _ptr(d_n), cap, i3(subm_ksize) if Ks else None,
                                i3(subm_dilation) if Ks else None, _ptr(subm_pair), _ptr(subm_cnt), d_st, _ptr(ws), wsb,
                                _stream(indices)), "spx_conv_rulebook")

    def with_subm(rb):
        if Ks:
            n = rb.n_out
            nxt = Rulebook(subm_pair, cap, n, n, Ks, True, rb.out_indices, out_shape, out_shape, cnt=subm_cnt,
                           ksize=list(subm_ksize), stride=[1, 1, 1], padding=[k // 2 for k in subm_ksize],
                           dilation=list(subm_dilation))
            nxt.d_n_in = nxt.d_n_out = rb.d_n_out
            rb.subm_next = nxt
        return rb
    if not sync:
        rb = Rulebook(pair_fwd, cap, n_in, cap, K, False, out_idx, out_shape, spatial_shape, pair_bwd=pair_bwd, cnt=cnt,
                      ksize=list(ksize), stride=list(stride), padding=list(padding), dilation=list(dilation))
        rb.d_n_in, rb.d_n_out = d_n_in, d_n
        return with_subm(rb)
    if sync == "later":
        # The row count is read back asynchronously; .finish() waits for it.  A caller that launches this table one stage
        # early (before it queues the previous stage's kernels) finds the count already there: the host never waits on an
        # empty GPU (pcdet_amd/models/backbones_3d/spconv_backbone.py).
        return PendingRulebook(d_n, lambda n_out: with_subm(Rulebook(
            pair_fwd, cap, n_in, n_out, K, False, out_idx[:n_out], out_shape, spatial_shape, pair_bwd=pair_bwd, cnt=cnt,
            ksize=list(ksize), stride=list(stride), padding=list(padding), dilation=list(dilation))))
    n_out = int(d_n.item())
    rb = Rulebook(pair_fwd, cap, n_in, n_out, K, False, out_idx[:n_out], out_shape, spatial_shape, pair_bwd=pair_bwd,
                  cnt=cnt, ksize=list(ksize), stride=list(stride), padding=list(padding), dilation=list(dilation))
    return with_subm(rb)


class PendingRulebook(object):
    """A strided rule table whose kernels are queued and whose row count is on its way to the host."""

    _pool, _next = [], 0             # pinned host words, reused round-robin (pinning memory costs far more than the copy)

    def __init__(self, d_n, make):
        cls = PendingRulebook
        if len(cls._pool) < 16:
            cls._pool.append(torch.empty((1,), dtype=torch.int64).pin_memory())
            self._host = cls._pool[-1]
        else:
            self._host = cls._pool[cls._next % 16]
            cls._next += 1
        self._make = make
        self._host.copy_(d_n, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record(torch.cuda.current_stream(d_n.device))
        self._d_n = d_n                          # keeps the source of the copy alive

    def finish(self):
        self._event.synchronize()
        return self._make(int(self._host[0]))


# ------------------------------------------------------------------------------------------- arithmetic

def pack_weight(weight, mode, out=None):
    """weight [Cout, kz, ky, kx, Cin] (or [Cout, K, Cin]) -> MFMA operand order.  mode 0 fwd, 1 dgrad, 2 both.  out: a buffer
    of a previous call with the same weight shape and mode, refilled in place."""
    _need_gpu(weight)
    lib = _lib.load()
    w = weight.detach().contiguous().float()
    cout, cin = w.shape[0], w.shape[-1]
    K = w.numel() // (cout * cin)
    n = (2 if mode == 2 else 1) * K * cin * cout
    packed = out if out is not None else torch.empty((n,), dtype=torch.float32, device=w.device)
    assert packed.numel() == n and packed.is_contiguous() and packed.device == w.device
    check(lib.spx_pack_weight(_ptr(w), cout, K, cin, mode, _ptr(packed), _stream(w)), "spx_pack_weight")
    return packed                # mode 2: forward operand in the first half, dgrad operand in the second


class PackedBank(object):
    """Both MFMA operand orders of a fixed list of contiguous weights, re-packed by ONE launch (spx_pack_weight_batched).
    halves(i) -> (forward operand, dgrad operand) views of weight i."""

    def __init__(self, weights):
        self.weights = list(weights)
        dev = self.weights[0].device
        sizes = [w.numel() for w in self.weights]
        self.flat = torch.empty((2 * sum(sizes),), dtype=torch.float32, device=dev)
        rows, off, blk = [], 0, 0
        self.views = []
        for w, n in zip(self.weights, sizes):
            cout, cin = w.shape[0], w.shape[-1]
            rows.append([w.data_ptr(), self.flat.data_ptr() + 4 * off, cout, n // (cout * cin), cin, blk])
            self.views.append((self.flat[off:off + n], self.flat[off + n:off + 2 * n]))
            off += 2 * n
            blk += (2 * n + 255) // 256
        self.blocks = blk
        self.desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.ptrs = tuple(w.data_ptr() for w in self.weights)

    def valid_for(self, weights):
        return len(weights) == len(self.weights) and tuple(w.data_ptr() for w in weights) == self.ptrs

    def repack(self):
        check(_lib.load().spx_pack_weight_batched(_ptr(self.desc), len(self.weights), self.blocks, _stream(self.flat)),
              "spx_pack_weight_batched")


def conv_gemm(src, w_packed, c_dst, kvol, pair, ld, n_dst, flip_k=False, scale=None, shift=None, relu=False,
              d_n_dst=None):
    """d_n_dst: optional device int64[1] live destination-row count (n_dst is then the launch capacity; rows beyond the
    live count are not written)."""
    _need_gpu(src, w_packed, pair)
    lib = _lib.load()
    src = src.contiguous()
    assert src.dtype == torch.float32
    dst = torch.empty((n_dst, c_dst), dtype=torch.float32, device=src.device)
    if n_dst == 0:
        return dst
    if src.shape[0] == 0:
        return dst.zero_()
    check(lib.spx_conv_gemm(_ptr(src), src.shape[1], _ptr(w_packed), c_dst, kvol, int(bool(flip_k)), _ptr(pair), ld,
                            n_dst, _ptr(d_n_dst), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(dst), _stream(src)),
          "spx_conv_gemm")
    return dst


_BALANCED_SHAPES = {(64, 64)}   # measured: 32x32 slower, 32x64 / 64x32 even (plan cost included)
if os.environ.get("SPX_CONV_BALANCED_SHAPES"):   # dev knob: "32x32,32x64,64x32,64x64"
    _BALANCED_SHAPES = {tuple(int(v) for v in t.split("x")) for t in os.environ["SPX_CONV_BALANCED_SHAPES"].split(",")}
# below this many destination rows the plan / fix-up launches cost more than the balanced schedule saves
_BALANCED_MIN_ROWS = int(os.environ.get("SPX_CONV_BALANCED_MIN_ROWS", "24000"))
_BALANCED = os.environ.get("SPX_CONV_BALANCED", "1") != "0"
_GROUPED = os.environ.get("SPX_CONV_GROUPED", "1") != "0"       # dev knob: rows grouped by offset mask (csrc/conv_group.hip)


def conv_plan(pair, ld, kvol, n_dst, d_n_dst=None):
    """Work plan of the balanced schedule for one rule table (include/spx.h: spx_conv_plan); int32 device tensor."""
    _need_gpu(pair)
    lib = _lib.load()
    plan = torch.empty((lib.spx_conv_plan_bytes(n_dst) // 4,), dtype=torch.int32, device=pair.device)
    check(lib.spx_conv_plan(_ptr(pair), ld, kvol, n_dst, _ptr(d_n_dst), _ptr(plan), _stream(pair)), "spx_conv_plan")
    return plan


def conv_group(pair, ld, kvol, n_dst, d_n_dst=None):
    """Rows of a rule table grouped by offset mask (include/spx.h: spx_conv_group) -> perm [n_dst] int32 and the table
    in that order [kvol, n_dst] int32."""
    _need_gpu(pair)
    lib = _lib.load()
    perm = torch.empty((n_dst,), dtype=torch.int32, device=pair.device)
    grouped = torch.empty((kvol, n_dst), dtype=torch.int32, device=pair.device)
    wsb = lib.spx_conv_group_ws_bytes(n_dst)
    ws = workspace(pair.device, wsb)
    check(lib.spx_conv_group(_ptr(pair), ld, kvol, n_dst, _ptr(d_n_dst), _ptr(perm), _ptr(grouped), _ptr(ws), wsb,
                             _stream(pair)), "spx_conv_group")
    return perm, grouped


def grouped_plan_for(rb, pair, ld, kvol, n_dst, d_n_dst=None):
    """(perm, grouped table, plan over it) of rule table `pair`, built once per Rulebook and table."""
    key = ("g", pair.data_ptr(), int(n_dst))
    hit = rb._plans.get(key)
    if hit is None:
        perm, grouped = conv_group(pair, ld, kvol, n_dst, d_n_dst)
        hit = rb._plans[key] = (perm, grouped, conv_plan(grouped, n_dst, kvol, n_dst, d_n_dst))
    return hit


def plan_for(rb, pair, ld, kvol, n_dst, d_n_dst=None):
    """Plan of rule table `pair`, built once per Rulebook and table (forward, dgrad and sibling layers share it)."""
    key = (pair.data_ptr(), int(n_dst))
    plan = rb._plans.get(key)
    if plan is None:
        plan = rb._plans[key] = conv_plan(pair, ld, kvol, n_dst, d_n_dst)
    return plan


def grouped_ok(rb, kvol):
    """Grouping the rows costs one sort per rule table: worth it for submanifold tables (forward and dgrad of two
    layers read the same one), not for the single-use tables of strided convolutions."""
    return _GROUPED and rb.subm and kvol <= 30


def balanced_ok(c_src, c_dst, n_dst, rb=None, pair=None):
    if not _BALANCED or n_dst < _BALANCED_MIN_ROWS:
        return False
    if (c_src, c_dst) in _BALANCED_SHAPES:
        return True
    # 128 -> 128: the two-half kernel (k_conv_mfma_pbl2) beats one tile per wave on tables whose 64-row super-tiles
    # really share their offsets — the forward table of the BEV entry conv (313 -> 209 us); on its backward table, where
    # every row has 9 of the 18 offsets, it loses (185 vs 165 us).  The caller marks the rulebook.
    return (c_src, c_dst) == (128, 128) and rb is not None and getattr(rb, "wide_balanced_fwd", False) and pair is rb.pair


def conv_gemm_balanced(src, w_packed, c_dst, kvol, pair, ld, n_dst, plan, flip_k=False, scale=None, shift=None,
                       relu=False, d_n_dst=None, perm=None):
    """conv_gemm under the MFMA-work-balanced persistent schedule (`plan` from conv_plan on the same table / n_dst)."""
    _need_gpu(src, w_packed, pair, plan)
    lib = _lib.load()
    src = src.contiguous()
    assert src.dtype == torch.float32 and n_dst > 0 and src.shape[0] > 0
    dst = torch.empty((n_dst, c_dst), dtype=torch.float32, device=src.device)
    wsb = lib.spx_conv_gemm_balanced_ws_bytes(c_dst, n_dst)
    ws = workspace(src.device, wsb)
    check(lib.spx_conv_gemm_balanced(_ptr(src), src.shape[1], _ptr(w_packed), c_dst, kvol, int(bool(flip_k)), _ptr(pair),
                                     ld, n_dst, _ptr(d_n_dst), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(plan),
                                     _ptr(perm), _ptr(dst), _ptr(ws), wsb, _stream(src)), "spx_conv_gemm_balanced")
    return dst


_RING = os.environ.get("SPX_CONV_RING", "1") != "0"              # dev knob: round-3 schedule (csrc/conv_ring.hip)
# 32x32 / 32x64 / 64x32 are built too and 15-25 % faster than the one-tile-per-wave kernels launch by launch (kbench --ring),
# but the training step gets SLOWER with them (299.9 vs 303.7 frames/s, same call): four more plans on the index stream and
# 1024-thread persistent workgroups that have to wait for whole CUs beside the side-stream weight-gradient kernels
_RING_SHAPES = {(64, 64)}
if os.environ.get("SPX_CONV_RING_SHAPES"):                       # dev knob: "32x32,32x64,64x32,64x64"
    _RING_SHAPES = {tuple(int(v) for v in t.split("x")) for t in os.environ["SPX_CONV_RING_SHAPES"].split(",")}
_RING_MIN_ROWS = int(os.environ.get("SPX_CONV_RING_MIN_ROWS", "24000"))


def ring_ok(c_src, c_dst, n_dst, kvol):
    return _RING and (c_src, c_dst) in _RING_SHAPES and n_dst >= _RING_MIN_ROWS and kvol <= 31


def conv_ring_plan(pair, ld, kvol, n_dst, d_n_dst=None):
    """Tile -> wave assignment of the ring schedule for one rule table (include/spx.h: spx_conv_ring_plan)."""
    _need_gpu(pair)
    lib = _lib.load()
    plan = torch.empty((lib.spx_conv_ring_plan_bytes(n_dst) // 4,), dtype=torch.int32, device=pair.device)
    check(lib.spx_conv_ring_plan(_ptr(pair), ld, kvol, n_dst, _ptr(d_n_dst), _ptr(plan), _stream(pair)),
          "spx_conv_ring_plan")
    return plan


def ring_plan_for(rb, pair, ld, kvol, n_dst, d_n_dst=None):
    """(perm, table, ring plan) of rule table `pair`, built once per Rulebook and table: over the grouped rows for submanifold
    tables (perm != None), over the plain table otherwise."""
    key = ("r", pair.data_ptr(), int(n_dst))
    hit = rb._plans.get(key)
    if hit is None:
        if grouped_ok(rb, kvol):
            gkey = ("g", pair.data_ptr(), int(n_dst))
            ghit = rb._plans.get(gkey)
            if ghit is not None:
                perm, grouped = ghit[0], ghit[1]
            else:
                perm, grouped = conv_group(pair, ld, kvol, n_dst, d_n_dst)
            hit = (perm, grouped, n_dst, conv_ring_plan(grouped, n_dst, kvol, n_dst, d_n_dst))
        else:
            hit = (None, pair, ld, conv_ring_plan(pair, ld, kvol, n_dst, d_n_dst))
        rb._plans[key] = hit
    return hit


def conv_gemm_ring(src, w_packed, c_dst, kvol, pair, ld, n_dst, plan, flip_k=False, scale=None, shift=None, relu=False,
                   d_n_dst=None, perm=None, want_stats=False):
    """conv_gemm under the ring schedule (`plan` from conv_ring_plan on the same table / n_dst).  want_stats: also return the
    per-workgroup column sums [rows, 2, c_dst] of the written values and their squares."""
    _need_gpu(src, w_packed, pair, plan)
    lib = _lib.load()
    src = src.contiguous()
    assert src.dtype == torch.float32 and n_dst > 0 and src.shape[0] > 0
    dst = torch.empty((n_dst, c_dst), dtype=torch.float32, device=src.device)
    stats = None
    if want_stats:
        stats = torch.empty((lib.spx_conv_ring_stat_rows(), 2, c_dst), dtype=torch.float32, device=src.device)
    check(lib.spx_conv_gemm_ring(_ptr(src), src.shape[0], src.shape[1], _ptr(w_packed), c_dst, kvol, int(bool(flip_k)),
                                 _ptr(pair), ld, n_dst, _ptr(d_n_dst), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(plan),
                                 _ptr(perm), _ptr(dst), _ptr(stats), _ptr(status_word(src.device)), _stream(src)),
          "spx_conv_gemm_ring")
    return (dst, stats) if want_stats else dst


def wgrad_counts(pair, ld, kvol, n_out, d_n_out=None):
    """Pair counts of a rule table for conv_wgrad's work split (include/spx.h: spx_conv_wgrad_counts); int32 device tensor."""
    _need_gpu(pair)
    lib = _lib.load()
    counts = torch.empty((lib.spx_conv_wgrad_counts_bytes(kvol, n_out) // 4,), dtype=torch.int32, device=pair.device)
    check(lib.spx_conv_wgrad_counts(_ptr(pair), ld, kvol, n_out, _ptr(d_n_out), _ptr(counts), _stream(pair)),
          "spx_conv_wgrad_counts")
    return counts


def wgrad_counts_for(rb, pair, ld, kvol, n_out, d_n_out=None):
    """Counts of rule table `pair`, computed once per Rulebook (shared by the layers that use the table)."""
    key = ("wc", pair.data_ptr(), int(n_out))
    hit = rb._plans.get(key)
    if hit is None:
        hit = rb._plans[key] = wgrad_counts(pair, ld, kvol, n_out, d_n_out)
    return hit


def conv_wgrad(feat_in, dout, pair, ld, n_out, wshape, d_n_out=None, counts=None):
    """d_n_out: optional device int64[1] live output-row count (n_out is then the capacity); counts: wgrad_counts of the
    table (computed inside the call when None)."""
    _need_gpu(feat_in, dout, pair)
    lib = _lib.load()
    feat_in = feat_in.contiguous()
    dout = dout.contiguous()
    cout, cin = wshape[0], wshape[-1]
    K = 1
    for s in wshape[1:-1]:
        K *= int(s)
    dw = torch.empty(tuple(wshape), dtype=torch.float32, device=dout.device)
    if feat_in.shape[0] == 0 or n_out == 0:
        return dw.zero_()
    wsb = lib.spx_conv_wgrad_ws_bytes(cin, cout, K, n_out)
    ws = workspace(dout.device, wsb)
    check(lib.spx_conv_wgrad(_ptr(feat_in), cin, _ptr(dout), cout, K, _ptr(pair), ld, n_out, _ptr(d_n_out), _ptr(counts),
                             _ptr(dw), _ptr(ws),
                             wsb, _stream(dout)), "spx_conv_wgrad")
    return dw


# ------------------------------------------------------------------------------------------- densify

def densify(features, indices, batch_size, spatial_shape, channels_last=False, d_n=None):
    """[N,C] -> logical [B,C,D,H,W].  channels_last=True stores it as [B,H,W,C,D] (see include/spx.h §5).
    d_n: optional device int64[1] live row count."""
    _need_gpu(features, indices)
    lib = _lib.load()
    features = features.contiguous()
    n, c = features.shape
    d, h, w = [int(s) for s in spatial_shape]
    dev = features.device
    if channels_last:
        dense = torch.zeros((batch_size, h, w, c, d), dtype=torch.float32, device=dev)
    else:
        dense = torch.zeros((batch_size, c, d, h, w), dtype=torch.float32, device=dev)
    check(lib.spx_densify(_ptr(features), _ptr(indices), n, _ptr(d_n), c, batch_size, i3(spatial_shape),
                          1 if channels_last else 0, _ptr(dense), _stream(features)), "spx_densify")
    return dense.permute(0, 3, 4, 1, 2) if channels_last else dense


def densify_bwd(ddense, indices, batch_size, spatial_shape, channels_last=False, d_n=None):
    _need_gpu(ddense, indices)
    lib = _lib.load()
    n = indices.shape[0]
    c = ddense.shape[1]
    if channels_last:
        dd = ddense.permute(0, 3, 4, 1, 2).contiguous()  # -> [B,H,W,C,D] storage
    else:
        dd = ddense.contiguous()
    dfeat = torch.empty((n, c), dtype=torch.float32, device=ddense.device)
    check(lib.spx_densify_bwd(_ptr(dd), _ptr(indices), n, _ptr(d_n), c, batch_size, i3(spatial_shape),
                              1 if channels_last else 0, _ptr(dfeat), _stream(ddense)), "spx_densify_bwd")
    return dfeat


# ------------------------------------------------------------------------------------------- rotated BEV IoU / NMS

def boxes_iou_bev(boxes_a, boxes_b, overlap_only=False):
    """(N,7),(M,7) [x,y,z,dx,dy,dz,heading] -> (N,M) BEV IoU (or intersection area)."""
    _need_gpu(boxes_a, boxes_b)
    lib = _lib.load()
    a = boxes_a[:, :7].contiguous().float()
    b = boxes_b[:, :7].contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib.spx_boxes_iou_bev(_ptr(a), a.shape[0], _ptr(b), b.shape[0], int(bool(overlap_only)), _ptr(out), _stream(a)),
          "spx_boxes_iou_bev")
    return out


def nms_bev(boxes_sorted, thresh, axis_aligned=False):
    """boxes (N,7) already sorted by descending score -> (keep positions int64 [N] on device, count tensor int64[1]).
    No host sync here; the caller slices keep[:count]."""
    _need_gpu(boxes_sorted)
    lib = _lib.load()
    b = boxes_sorted[:, :7].contiguous().float()
    n = b.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    cnt = torch.zeros((1,), dtype=torch.int64, device=b.device)
    wsb = lib.spx_nms_ws_bytes(n)
    ws = workspace(b.device, wsb)
    check(lib.spx_nms_bev(_ptr(b), n, float(thresh), int(bool(axis_aligned)), _ptr(keep), _ptr(cnt), _ptr(ws), wsb,
                          _stream(b)), "spx_nms_bev")
    return keep, cnt


# ------------------------------------------------------------------------------------------- anchor target assignment

def assign_targets(anchors, per_location, gt_boxes, set_class, n_classes, matched, unmatched):
    """anchors [n_sets, A, 7] fp32, gt_boxes [B, M, 8] fp32, set_class int32[n_sets], matched/unmatched fp32[n_sets]
    (all on the GPU) -> labels int32 [B, n_sets*A], targets fp32 [B, n_sets*A, 7], weights fp32 [B, n_sets*A]
    in the head's (y, x, set, within-location) anchor order.  No host sync."""
    _need_gpu(anchors, gt_boxes, set_class, matched, unmatched)
    lib = _lib.load()
    n_sets, a, _ = anchors.shape
    gt = gt_boxes.contiguous().float()
    b, m = gt.shape[0], gt.shape[1]
    dev = gt.device
    labels = torch.empty((b, n_sets * a), dtype=torch.int32, device=dev)
    targets = torch.empty((b, n_sets * a, 7), dtype=torch.float32, device=dev)
    weights = torch.empty((b, n_sets * a), dtype=torch.float32, device=dev)
    wsb = lib.spx_assign_targets_ws_bytes(b, n_sets, m)
    ws = workspace(dev, wsb)
    check(lib.spx_assign_targets(_ptr(anchors), n_sets, a, int(per_location), _ptr(gt), b, m, _ptr(set_class),
                                 int(n_classes), _ptr(matched), _ptr(unmatched), _ptr(labels), _ptr(targets),
                                 _ptr(weights), _ptr(ws), wsb, _stream(gt)), "spx_assign_targets")
    return labels, targets, weights


# ------------------------------------------------------------------------------------------- anchor-head losses

def anchor_loss(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, dir_offset, cls_weight, loc_weight,
                dir_weight, beta=1.0 / 9.0, alpha=0.25):
    """cls_preds [B,A,NC], box_preds [B,A,7], dir_preds [B,A,NB] or None, labels int32 [B,A], reg_targets [B,A,7],
    anchors [A,7] -> (losses fp32[3] = weighted cls/loc/dir, dcls, dbox, ddir).  No host sync."""
    _need_gpu(cls_preds, box_preds, labels, reg_targets, anchors)
    lib = _lib.load()
    cls_preds, box_preds = cls_preds.contiguous().float(), box_preds.contiguous().float()
    dir_c = None if dir_preds is None else dir_preds.contiguous().float()
    labels = labels.contiguous()
    assert labels.dtype == torch.int32
    reg_targets, anchors = reg_targets.contiguous().float(), anchors.contiguous().float()
    b, a, nc = cls_preds.shape
    nb = 0 if dir_c is None else dir_c.shape[2]
    dev = cls_preds.device
    losses = torch.zeros((3,), dtype=torch.float32, device=dev)
    dcls, dbox = torch.empty_like(cls_preds), torch.empty_like(box_preds)
    ddir = None if dir_c is None else torch.empty_like(dir_c)
    wsb = lib.spx_anchor_loss_ws_bytes(b, a)
    ws = workspace(dev, wsb)
    check(lib.spx_anchor_loss(_ptr(cls_preds), _ptr(box_preds), _ptr(dir_c), _ptr(labels), _ptr(reg_targets),
                              _ptr(anchors), b, a, nc, nb, float(dir_offset), float(cls_weight), float(loc_weight),
                              float(dir_weight), float(beta), float(alpha), _ptr(losses), _ptr(dcls), _ptr(dbox),
                              _ptr(ddir), _ptr(ws), wsb, _stream(cls_preds)), "spx_anchor_loss")
    return losses, dcls, dbox, ddir


# ------------------------------------------------------------------------------------------- BatchNorm1d (+ReLU), training

def bn_relu_supported(c):
    return c % 4 == 0 and 1024 % c == 0


def _row_view_ok(t, n, c):
    """[n, c] float32 view whose rows are contiguous and 16-byte aligned at a row stride that is a multiple of 4."""
    return (t.dim() == 2 and tuple(t.shape) == (n, c) and t.dtype == torch.float32 and (n <= 1 or t.stride(0) % 4 == 0)
            and (c == 1 or t.stride(1) == 1) and t.stride(0) >= c and t.data_ptr() % 16 == 0)


def bn_relu_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual=None, out=None,
                num_batches_tracked=None, d_n=None):
    """Training-mode BatchNorm1d over the rows of x [N, C] (+ residual) (+ReLU).  Updates running_mean / running_var
    (and num_batches_tracked, when given) in place.  `out`: optional [N, C] view with its own row stride (a channel
    slice of a wider matrix) that receives y.  Returns y, save_mean, save_invstd."""
    _need_gpu(x, gamma, beta)
    lib = _lib.load()
    x = x.contiguous()
    n, c = x.shape
    if residual is not None:
        residual = residual.contiguous()
        if residual.shape != x.shape or residual.dtype != x.dtype or residual.device != x.device:
            raise ValueError("residual must match x: %s vs %s" % (tuple(residual.shape), tuple(x.shape)))
    if out is None:
        y = torch.empty_like(x)
    else:
        if not (_row_view_ok(out, n, c) and out.device == x.device):
            raise ValueError("out must be a float32 [N, C] row view (row stride a multiple of 4, 16-byte aligned)")
        y = out
    y_ld = y.stride(0) if n > 1 else c
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
    wsb = lib.spx_bn_relu_ws_bytes(c)
    ws = workspace(x.device, wsb)
    check(lib.spx_bn_add_relu_fwd(_ptr(x), _ptr(residual), n, _ptr(d_n), c, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                  _ptr(running_var), _ptr(num_batches_tracked), float(momentum), float(eps),
                                  int(bool(relu)), _ptr(y), y_ld, _ptr(mean), _ptr(invstd), _ptr(ws), wsb, _stream(x)),
          "spx_bn_add_relu_fwd")
    return y, mean, invstd


def bn_relu_fwd_from_sums(x, partial, gamma, beta, running_mean, running_var, momentum, eps, relu, num_batches_tracked=None,
                          d_n=None):
    """bn_relu_fwd for rows x [N, C] whose per-block sums [rows, 2, C] the producing kernel already took (conv2d_wino with
    stats=True, conv_gemm_ring with want_stats=True): no statistics pass over x.  d_n: device-side live row count of a
    static-capacity x.  Returns y, save_mean, save_invstd."""
    _need_gpu(x, partial, gamma, beta)
    lib = _lib.load()
    x = x.contiguous()
    n, c = x.shape
    assert partial.dim() == 3 and partial.shape[1] == 2 and partial.shape[2] == c and partial.is_contiguous()
    y = torch.empty_like(x)
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
    check(lib.spx_bn_relu_fwd_from_sums(_ptr(x), n, _ptr(d_n), c, _ptr(partial), int(partial.shape[0]), _ptr(gamma), _ptr(beta),
                                        _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), float(momentum),
                                        float(eps), int(bool(relu)), _ptr(y), c, _ptr(mean), _ptr(invstd), _stream(x)),
          "spx_bn_relu_fwd_from_sums")
    return y, mean, invstd


def bn_apply(x, mean, invstd, gamma, beta, relu, residual=None, out=None, d_n=None):
    """Inference-mode BatchNorm (+ residual) (+ReLU) over the rows of x [N, C] with given statistics, one pass.  `out`:
    optional [N, C] row view with its own stride (a channel slice of a wider matrix)."""
    _need_gpu(x, mean, invstd, gamma, beta)
    lib = _lib.load()
    x = x.contiguous()
    n, c = x.shape
    if out is None:
        y = torch.empty_like(x)
    else:
        if not (_row_view_ok(out, n, c) and out.device == x.device):
            raise ValueError("out must be a float32 [N, C] row view (row stride a multiple of 4, 16-byte aligned)")
        y = out
    y_ld = y.stride(0) if n > 1 else c
    if residual is not None:
        residual = residual.contiguous()
    check(lib.spx_bn_apply(_ptr(x), _ptr(residual), n, _ptr(d_n), c, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
                           int(bool(relu)), _ptr(y), y_ld, _stream(x)), "spx_bn_apply")
    return y


def wino_ok(cin, cout):
    """Channel counts spx_conv2d_wino takes (else the caller keeps the vendor convolution)."""
    return cin % 32 == 0 and cout % 128 == 0


def wino_weight(weight, flip=False, out=None):
    """Transformed weight image of a [Cout, Cin, 3, 3] filter (any strides) for conv2d_wino.  flip: the image of the DATA
    GRADIENT's filter (taps rotated, channel roles swapped) — conv2d_wino(dy, wino_weight(w, flip=True)) = dx.
    out: a buffer of a previous call for the same shape to refill instead of allocating."""
    _need_gpu(weight)
    assert weight.dim() == 4 and weight.shape[2] == 3 and weight.shape[3] == 3 and weight.dtype == torch.float32
    lib = _lib.load()
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    k_in, k_out = (cout, cin) if flip else (cin, cout)
    nfl = lib.spx_wino_weight_floats(k_in, k_out)
    if out is not None and out.numel() == nfl and out.device == weight.device and out.dtype == torch.float32:
        u = out
    else:
        u = torch.empty((nfl,), dtype=torch.float32, device=weight.device)
    so, si, sa, sb = weight.stride()
    check(lib.spx_wino_weight(_ptr(weight), so, si, sa, sb, k_in, k_out, int(bool(flip)), _ptr(u), _stream(weight)),
          "spx_wino_weight")
    return u


def conv2d_wino(x, u, cout, scale=None, shift=None, relu=False, out=None, stats=False):
    """3x3 / stride 1 / pad 1 convolution of a channels-last map.  x: [N, Cin, H, W] tensor in channels_last memory format
    (or any [N, Cin, H, W] view whose pixels are dense rows: stride (H*W*ld, 1, W*ld, ld)); u: wino_weight image;
    returns [N, cout, H, W] channels_last (or writes `out`, same layout rule).  Optional epilogue relu?(y*scale + shift).
    stats=True: also returns the [rows, 2, cout] per-tile-block sums of y and y*y (bn_relu_fwd_from_sums takes them)."""
    _need_gpu(x, u)
    lib = _lib.load()
    n, cin, h, w = (int(v) for v in x.shape)
    x_ld = _cl_ld(x)
    if x_ld is None:
        x = x.contiguous(memory_format=torch.channels_last)
        x_ld = _cl_ld(x)
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    y_ld = _cl_ld(out)
    if y_ld is None or tuple(out.shape) != (n, cout, h, w):
        raise ValueError("out must be a float32 [N, Cout, H, W] channels-last map")
    if u.numel() != lib.spx_wino_weight_floats(cin, cout):
        raise ValueError("weight image does not match Cin=%d, Cout=%d" % (cin, cout))
    part = None
    if stats:
        part = torch.empty((lib.spx_wino_stat_rows(n, h, w), 2, cout), dtype=torch.float32, device=x.device)
    check(lib.spx_conv2d_wino(_ptr(x), x_ld, _ptr(u), n, h, w, cin, cout, _ptr(scale), _ptr(shift), int(bool(relu)),
                              _ptr(out), y_ld, _ptr(part), _stream(x)), "spx_conv2d_wino")
    return (out, part) if stats else out


def wino_wgrad_ok(cin, cout, w):
    """Shapes spx_conv2d_wino_wgrad takes (else the caller keeps the vendor weight-gradient kernel)."""
    return cin % 128 == 0 and cout % 128 == 0 and (w + 1) // 2 >= 8


def conv2d_wino_wgrad(x, dy, like):
    """Weight gradient of conv2d(x, w, padding=1) for a [Cout, Cin, 3, 3] weight, written with the strides of `like`.
    x [N, Cin, H, W], dy [N, Cout, H, W]: channels-last maps (see conv2d_wino)."""
    _need_gpu(x, dy)
    lib = _lib.load()
    n, cin, h, w = (int(v) for v in x.shape)
    cout = int(dy.shape[1])
    x_ld, dy_ld = _cl_ld(x), _cl_ld(dy)
    if x_ld is None:
        x = x.contiguous(memory_format=torch.channels_last)
        x_ld = _cl_ld(x)
    if dy_ld is None:
        dy = dy.contiguous(memory_format=torch.channels_last)
        dy_ld = _cl_ld(dy)
    assert tuple(like.shape) == (cout, cin, 3, 3) and tuple(dy.shape) == (n, cout, h, w)
    dw = torch.empty_like(like)
    so, si, sa, sb = dw.stride()
    wsb = lib.spx_wino_wgrad_ws_bytes(cin, cout)
    ws = workspace(x.device, wsb)
    check(lib.spx_conv2d_wino_wgrad(_ptr(x), x_ld, _ptr(dy), dy_ld, n, h, w, cin, cout, _ptr(dw), so, si, sa, sb, _ptr(ws), wsb,
                                    _stream(x)), "spx_conv2d_wino_wgrad")
    return dw


def _cl_ld(t):
    """Pixel pitch (floats) of a [N, C, H, W] float32 map whose memory is [N, H, W] rows of >= C channels, else None."""
    if t.dtype != torch.float32 or t.dim() != 4:
        return None
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    ld = sw if w > 1 else (sh if h > 1 else c)
    if c > 1 and sc != 1:
        return None
    if ld < c or ld % 4 or t.data_ptr() % 16:
        return None
    if (w > 1 and sw != ld) or (h > 1 and sh != w * ld) or (n > 1 and sn != h * w * ld):
        return None
    return int(ld)


def bn_relu_bwd(x, dy, gamma, beta, mean, invstd, relu, residual=None, d_n=None):
    """Backward of bn_relu_fwd; the ReLU mask is recomputed from x (and the residual) inside the kernels (y is not
    read).  dy may be a row view with its own stride (a channel slice of a wider gradient).  Returns dx, dgamma, dbeta
    and, with a residual, dresidual."""
    _need_gpu(x, dy)
    lib = _lib.load()
    n, c = x.shape
    if not _row_view_ok(dy, n, c):
        dy = dy.contiguous()
    dy_ld = dy.stride(0) if n > 1 else c
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if residual is not None else None
    dgamma = torch.empty((c,), dtype=torch.float32, device=x.device)
    dbeta = torch.empty((c,), dtype=torch.float32, device=x.device)
    if n == 0:
        out = (dx, dgamma.zero_(), dbeta.zero_())
        return out if residual is None else out + (dres,)
    wsb = lib.spx_bn_relu_ws_bytes(c)
    ws = workspace(x.device, wsb)
    check(lib.spx_bn_add_relu_bwd(_ptr(x), _ptr(residual), _ptr(dy), dy_ld, n, _ptr(d_n), c, _ptr(gamma), _ptr(beta), _ptr(mean),
                                  _ptr(invstd), int(bool(relu)), _ptr(dx), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
                                  wsb, _stream(x)), "spx_bn_add_relu_bwd")
    return (dx, dgamma, dbeta) if residual is None else (dx, dgamma, dbeta, dres)


# ------------------------------------------------------------------------------------------- voxel query (row f-4)

def voxel_query(new_xyz, xyz, new_coords, point_indices, nsample, radius, ranges):
    """Raw kernel result of include/spx.h §10: idx [M, nsample] int32 (slot 0 = -1 for empty balls) and cnt [M]."""
    _need_gpu(new_xyz, xyz, new_coords, point_indices)
    lib = _lib.load()
    new_xyz, xyz = new_xyz.contiguous().float(), xyz.contiguous().float()
    new_coords, point_indices = new_coords.contiguous().int(), point_indices.contiguous().int()
    m = new_coords.shape[0]
    b, z, y, x = point_indices.shape
    idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
    cnt = torch.zeros((m,), dtype=torch.int32, device=xyz.device)
    check(lib.spx_voxel_query(_ptr(new_xyz), _ptr(xyz), _ptr(new_coords), _ptr(point_indices), m, b, i3([z, y, x]),
                              int(nsample), float(radius), i3(ranges), _ptr(idx), _ptr(cnt), _stream(xyz)),
          "spx_voxel_query")
    return idx, cnt


def voxel_query_dilated(new_xyz, xyz, new_coords, point_indices, nsample, former_radius, radius, ranges, strides):
    """Raw kernel result of spx_voxel_query_dilated: idx [M, nsample] int32, cnt_unique [M] (occupied cells scanned) and
    idx_cnt [M] (slots filled before padding)."""
    _need_gpu(new_xyz, xyz, new_coords, point_indices)
    lib = _lib.load()
    new_xyz, xyz = new_xyz.contiguous().float(), xyz.contiguous().float()
    new_coords, point_indices = new_coords.contiguous().int(), point_indices.contiguous().int()
    m = new_coords.shape[0]
    b, z, y, x = point_indices.shape
    idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
    cnt = torch.zeros((m,), dtype=torch.int32, device=xyz.device)
    filled = torch.zeros((m,), dtype=torch.int32, device=xyz.device)
    check(lib.spx_voxel_query_dilated(_ptr(new_xyz), _ptr(xyz), _ptr(new_coords), _ptr(point_indices), m, b,
                                      i3([z, y, x]), int(nsample), float(former_radius), float(radius), i3(ranges),
                                      i3(strides), _ptr(idx), _ptr(cnt), _ptr(filled), _stream(xyz)),
          "spx_voxel_query_dilated")
    return idx, cnt, filled
