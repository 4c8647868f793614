"""Index pipeline on its own HIP stream.

The rule tables of a sparse network depend on voxel INDICES only, never on features: level L+1's tables can be built
while level L's convolutions are still multiplying.  In static-capacity mode (SparseConvTensor.n_valid set: every row
count stays on the device, nothing is read back) `prebuild()` queues ALL tables of a stack — submanifold hash probes,
strided bitmap ranks, the row grouping and the work plans of the balanced schedule, forward and backward — on a side
stream right after the voxeliser; each table carries an event that its first consumer on the compute stream waits for.
The index kernels are small, latency-bound launches (a few waves per CU): they run in the shadow of the MFMA kernels
instead of in front of them.  Under hipGraph capture the fork / join becomes two parallel branches of the graph.

Replaces the lazy, in-line indice-pair builds of spconv.pytorch's modules (reference call sites
pcdet/models/backbones_3d/spconv_backbone.py:86-122); the tables are the same objects the modules would have built
themselves (`indice_dict[indice_key]`), so module code and results do not change.
"""
import os

import torch

from . import ops

_SIDE = {}


def side_stream(device):
    s = _SIDE.get(device.index)
    if s is None:
        s = _SIDE[device.index] = torch.cuda.Stream(device=device)
    return s


def _plans(rb, conv, backward):
    """Warm the plan caches of `rb` for `conv` exactly as spx.functional._conv will ask for them."""
    cin, cout, kvol = conv.in_channels, conv.out_channels, rb.kvol
    jobs = [(cin, cout, rb.pair, rb.ld, rb.n_out, rb.d_n_out)]
    if backward:
        if rb.subm:
            jobs.append((cout, cin, rb.pair, rb.ld, rb.n_in, rb.d_n_in))
        else:
            jobs.append((cout, cin, rb.pair_bwd, rb.pair_bwd.shape[1], rb.n_in, rb.d_n_in))
    if backward and rb.n_out > 0 and rb.n_in > 0:
        ops.wgrad_counts_for(rb, rb.pair, rb.ld, kvol, rb.n_out, rb.d_n_out)      # the weight gradient's work split
    for c_src, c_dst, pair, ld, n_dst, d_n in jobs:
        if n_dst > 0 and ops.ring_ok(c_src, c_dst, n_dst, kvol):
            ops.ring_plan_for(rb, pair, ld, kvol, n_dst, d_n)
            continue
        if n_dst <= 0 or not ops.balanced_ok(c_src, c_dst, n_dst, rb, pair):
            continue
        if ops.grouped_ok(rb, kvol):
            ops.grouped_plan_for(rb, pair, ld, kvol, n_dst, d_n)
        else:
            ops.plan_for(rb, pair, ld, kvol, n_dst, d_n)


def prebuild(x, convs, backward=None):
    """Queue every rule table the non-inverse SparseConvolution modules `convs` (execution order) will look up, starting
    from tensor `x`, on the side stream; no-op for exact-size tensors (their strided tables need a host read each).
    Returns the number of tables queued."""
    if x.n_valid is None or not x.features.is_cuda or x.indices.shape[0] == 0:
        return 0
    if backward is None:
        backward = torch.is_grad_enabled()
    dev = x.features.device
    main = torch.cuda.current_stream(dev)
    side = side_stream(dev)
    users = {}
    for m in convs:
        users.setdefault(m.indice_key, []).append(m)
    if torch.cuda.is_current_stream_capturing():
        # hipGraph replay (ROCm 7) cuts a captured graph into chains by a depth-first walk that follows, at every node, the child
        # captured FIRST; each chain runs on a stream of its own and those streams share a hardware queue, in which a chain whose
        # head waits for a late index kernel blocks every chain submitted behind it.  With the plain fork below the walk follows
        # the index chain, chops the compute chain at every table event, and the first convolution of the replay sat behind a
        # barrier until the LAST plan it shared a queue with was built (rocprofv3: conv_input at 430 us instead of ~110).  Fork from
        # an event instead and capture one compute-stream node before the index stream's first: the walk then takes the compute
        # chain in one piece and the index chain becomes the single side chain.
        fork = torch.cuda.Event()
        fork.record(main)
        ops.status_word(dev).add_(0)
        side.wait_event(fork)
    else:
        fork = None
        side.wait_stream(main)
    built = 0
    caps = x.static_caps or {}
    with torch.cuda.stream(side):
        idx, shape, d_n = x.indices, x.spatial_shape, x.n_valid
        pending = {}                       # submanifold tables that came out of a strided build (same rank bitmap)
        for pos, m in enumerate(convs):
            key = m.indice_key
            if key is None or m.inverse:
                continue
            rb = x.indice_dict.get(key)
            if rb is None:
                if m.subm:
                    rb = pending.pop(key, None)
                    if rb is None:
                        rb = ops.subm_rulebook(idx, x.batch_size, shape, m.kernel_size, m.dilation, d_n=d_n, unique=True)
                else:
                    # the first not-yet-built submanifold conv after this one works on this conv's output level
                    nxt = next((c for c in convs[pos + 1:] if not c.inverse and c.indice_key is not None
                                and c.indice_key not in x.indice_dict), None)
                    want = nxt is not None and nxt.subm
                    rb = ops.conv_rulebook(idx, x.batch_size, shape, m.kernel_size, m.stride, m.padding, m.dilation,
                                           d_n_in=d_n, cap=caps.get(key), sync=False,
                                           subm_ksize=nxt.kernel_size if want else None,
                                           subm_dilation=nxt.dilation if want else (1, 1, 1))
                    rb.in_indices = idx
                    if want:
                        pending[nxt.indice_key] = rb.subm_next
                for u in users[key]:
                    _plans(rb, u, backward)
                rb.ready = torch.cuda.Event()
                rb.ready.record(side)
                x.indice_dict[key] = rb
                built += 1
            if not m.subm:
                idx, shape, d_n = rb.out_indices, rb.out_shape, rb.d_n_out
    return built


def wait_ready(rb):
    """First consumer of a prebuilt table: make the compute stream wait for the side stream's event (once)."""
    ev = getattr(rb, "ready", None)
    if ev is not None:
        torch.cuda.current_stream(rb.pair.device).wait_event(ev)
        rb.ready = None
