"""autograd.Function wrappers: every forward AND backward below is a libspx HIP kernel (via spx.ops).

Replaces the autograd behaviour of spconv.pytorch convolutions / .dense() that the reference triggers with
loss.backward() (tools/train_utils/train_utils.py:53).
"""
import os
import weakref

import torch

from . import ops


def _packed(weight, mode):
    """MFMA-ordered copy of `weight`, cached until the parameter is modified in place (optimizer step, load).  The cache
    lives on the BASE tensor when `weight` is a view of a parameter (the BEV entry conv re-views its Conv2d weight per
    call; view and base share the version counter), keyed by the view's geometry.  For a weight that takes a gradient
    both operand orders are produced by ONE launch at forward time and the backward pass finds its half in the cache (the
    weights do not change in between) — decided on requires_grad alone: inside autograd.Function.forward grad mode is
    always off.  While a hipGraph is being captured the pack kernel is always recorded, so a replay never reads a stale
    copy."""
    if weight.is_cuda and torch.cuda.is_current_stream_capturing():
        if not torch.is_grad_enabled():           # an inference graph: the copy the warm-up passes cached, kept current by
            holder = weight._base if weight._base is not None else weight          # refresh_graph_constants()
            hit = (getattr(holder, "_spx_packed", None) or {}).get(
                (mode, weight.data_ptr(), tuple(weight.shape), tuple(weight.stride())))
            if hit is not None and hit[0] == weight._version:
                return _graph_constant(weight, hit[1], lambda w, out, mode=mode: ops.pack_weight(w, mode, out=out))
        return ops.pack_weight(weight, mode)      # a graph that trains (or no cached copy): the graph owns its buffers
    holder = weight._base if weight._base is not None else weight
    cache = getattr(holder, "_spx_packed", None)
    if cache is None:
        cache = {}
        try:
            holder._spx_packed = cache
        except AttributeError:
            return ops.pack_weight(weight, mode)
    geom = (weight.data_ptr(), tuple(weight.shape), tuple(weight.stride()))
    key = (mode,) + geom
    hit = cache.get(key)
    if hit is not None and hit[0] == weight._version:
        return hit[1]
    for k in [k for k, v in cache.items() if v[0] != weight._version]:
        del cache[k]
    if weight.is_cuda and (weight.requires_grad or holder.requires_grad):
        both = ops.pack_weight(weight, 2)
        half = both.numel() // 2
        cache[(0,) + geom] = (weight._version, both[:half])
        cache[(1,) + geom] = (weight._version, both[half:])
        return both[:half] if mode == 0 else both[half:]
    wp = ops.pack_weight(weight, mode)
    cache[key] = (weight._version, wp)
    return wp


_BANKS = weakref.WeakKeyDictionary()       # first conv module of a stack -> its PackedBank: released with the model
_FOLDED = weakref.WeakKeyDictionary()      # eval-mode BatchNorm module -> (key, scale, shift, invstd): released with the model


def folded_bn(bn, conv_bias=None):
    """(scale, shift, invstd) of an inference-mode BatchNorm on its running statistics, folded for a conv epilogue:
    y = conv * scale + shift, scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ conv_bias * scale).
    Cached per module: the five elementwise launches of the fold (rsqrt, mul, mul, sub, add) ran once per layer and FORWARD
    — 140 launches of ~4.7 us per cfg-5 forward, 0.6 ms of 4 (rocprofv3, round 3).  The cached tensors keep their storage: when
    a parameter or statistic was written since (its version counter moved, or it was replaced) they are recomputed IN PLACE, so
    a hipGraph that captured them picks the new values up (refresh_folded_bn)."""
    ts = (bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn.running_mean, bn.running_var, conv_bias)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts) + (float(bn.eps),)
    hit = _FOLDED.get(bn)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2], hit[3]
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var + bn.eps)
        scale = inv * bn.weight if bn.affine else inv.clone()
        shift = -bn.running_mean * scale
        if bn.affine:
            shift = shift + bn.bias
        if conv_bias is not None:
            shift = shift + conv_bias * scale
        if hit is not None and hit[1].shape == scale.shape and hit[1].device == scale.device:
            hit[1].copy_(scale)
            hit[2].copy_(shift)
            hit[3].copy_(inv)
            scale, shift, inv = hit[1], hit[2], hit[3]
        else:
            scale, shift, inv = scale.contiguous(), shift.contiguous(), inv.contiguous()
    _FOLDED[bn] = (key, scale, shift, inv, weakref.ref(conv_bias) if conv_bias is not None else None)
    return scale, shift, inv


_GRAPH_CONSTANTS = []      # [weakref(weight), version, buffer, refill(out)] of the constants inference graphs captured


def _graph_constant(weight, buf, refill):
    """Inference graphs only (capture in progress, no gradient recorded): let the graph read `buf` — the re-laid copy of `weight`
    (packed MFMA operand, Winograd image) that the eager warm-up passes before the capture left in the per-weight cache — instead
    of recording the kernel that rebuilds it: 13 pack + 10 transform launches of 3-9 us sat on the critical path of every
    captured forward (rocprofv3 timeline of a replay, round 3).  The buffer is registered; refresh_graph_constants() refills it
    IN PLACE when the weight was written since (GraphedDetector.__call__ runs it before every replay)."""
    holder = weight._base if weight._base is not None else weight      # a per-call view (BEV entry conv) dies with the call
    geom = (tuple(weight.shape), tuple(weight.stride()), weight.storage_offset())
    for ent in _GRAPH_CONSTANTS:           # a second graph over the same model reads the same buffers: one entry per buffer
        if ent[2].data_ptr() == buf.data_ptr() and ent[0]() is holder:
            return buf
    _GRAPH_CONSTANTS.append([weakref.ref(holder), weight._version, buf, refill, geom])
    return buf


def refresh_graph_constants():
    """Refill, in place, every constant an inference graph captured whose weight was written since (host-side version checks
    only while nothing changed); entries of weights that no longer exist are dropped."""
    keep = []
    for ent in _GRAPH_CONSTANTS:
        w = ent[0]()
        if w is None:
            continue
        if ent[1] != w._version:
            ent[3](w.detach().as_strided(*ent[4]), ent[2])
            ent[1] = w._version
        keep.append(ent)
    _GRAPH_CONSTANTS[:] = keep


def refresh_folded_bn(model):
    """Bring the cached folds of `model`'s BatchNorm modules up to date (in place) — call before replaying a hipGraph that was
    captured in inference mode when parameters may have been written since (GraphedDetector does).  Host-side version checks
    only while nothing changed."""
    for m in model.modules():
        hit = _FOLDED.get(m)
        if hit is not None:
            folded_bn(m, hit[4]() if hit[4] is not None else None)


def pack_all(convs):
    """Both operand orders of every weight of `convs` (SparseConvolution modules) in ONE launch, stored where _packed()
    looks (the per-parameter cache) — the optimizer changes all weights at once, so per step this replaces 2 launches per
    layer.  No-op while every cached copy is current, on the CPU, and under hipGraph capture."""
    ws = [m.weight for m in convs if m.weight.is_cuda and m.weight.is_contiguous() and m.weight.dtype == torch.float32]
    if not ws or torch.cuda.is_current_stream_capturing():
        return False
    stale = False
    for w in ws:
        c = getattr(w, "_spx_packed", None)
        geom = (w.data_ptr(), tuple(w.shape), tuple(w.stride()))
        hit = None if c is None else c.get((0,) + geom)
        if hit is None or hit[0] != w._version:
            stale = True
            break
    if not stale:
        return False
    key = convs[0]
    bank = _BANKS.get(key)
    if bank is None or not bank.valid_for(ws):
        bank = _BANKS[key] = ops.PackedBank(ws)
    bank.repack()
    for w, (fwd, bwd) in zip(ws, bank.views):
        geom = (w.data_ptr(), tuple(w.shape), tuple(w.stride()))
        try:
            w._spx_packed = {(0,) + geom: (w._version, fwd), (1,) + geom: (w._version, bwd)}
        except AttributeError:
            pass
    return True


_ASYNC_WGRAD = os.environ.get("SPX_ASYNC_WGRAD", "1") != "0"            # dev knob
_DEFER_JOIN = os.environ.get("SPX_WGRAD_DEFER_JOIN", "1") != "0"       # dev knob: 0 = join after every launch (what DDP gets)
_SIDE_STREAMS = {}


def _side_stream(device):
    s = _SIDE_STREAMS.get(device.index)
    if s is None:
        s = _SIDE_STREAMS[device.index] = torch.cuda.Stream(device=device)
    return s


_DEFERRED = set()          # id() of the weights whose gradient join was deferred in the running backward pass


def _join_after_backward(main, side, keep):
    """main waits for side when the running backward pass has finished (autograd engine callback, as DDP does for its
    buckets): whatever runs on `main` after loss.backward() — gradient clipping, the optimizer — sees the finished gradients.
    `keep`: the tensors the side kernels read; referenced until the join is enqueued, so that the allocator cannot hand their
    memory to main-stream kernels that would run BEFORE it (after it, stream order protects them).  One callback per deferred
    gradient (a stream wait is a few microseconds of host time): a once-per-pass flag would stay set for ever if a backward
    pass died with an exception before its callbacks ran, and every later pass would go unjoined."""
    def _join():
        main.wait_stream(side)
        keep.clear()
        _DEFERRED.clear()
    torch.autograd.Variable._execution_engine.queue_callback(_join)


def _off_critical_path(fn, reads, weight):
    """Weight gradients are not on the critical path of the backward pass — nothing reads them before the optimizer / the
    gradient all-reduce —, the data gradients are.  fn() (which launches the weight-gradient kernels and returns the
    gradient tensor) runs on a second stream: its kernels share the chip with whatever the main stream runs next (the
    memory-bound BatchNorm kernels, the small layers that do not fill it, the last partly empty round of a data gradient).
    `reads`: tensors fn's kernels read (produced on the main stream).  The main stream joins at the end of the backward pass
    when nothing reads the gradient before (see _nobody_reads_before_the_optimizer), else at once — the kernels then still
    overlap with the data gradient enqueued just before them."""
    dev = reads[0].device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        g = fn()
    g.record_stream(main)           # allocated in the side stream's pool, consumed (optimizer) and freed on the main stream
    # Deferring the join is only safe while AccumulateGrad STORES the tensor as it is (no kernel): a leaf without .grad and
    # hooks (see below), a gradient in the parameter's own layout (else the engine copies it into that layout on the main
    # stream), and the FIRST gradient of this weight in this pass — a weight used by two conv nodes gets its two gradients
    # summed by the engine's input buffer on the main stream as soon as the second one arrives.
    if (_nobody_reads_before_the_optimizer(weight) and id(weight) not in _DEFERRED
            and tuple(g.shape) == tuple(weight.shape) and g.stride() == weight.stride()):
        _DEFERRED.add(id(weight))
        _join_after_backward(main, side, list(reads))
    else:
        main.wait_stream(side)      # from here on stream order protects `reads` (and every earlier side-stream gradient)
    return g


def _nobody_reads_before_the_optimizer(weight):
    """True when the gradient returned for `weight` is only STORED during the backward pass (AccumulateGrad takes the tensor as
    it is: no kernel) — a leaf parameter without a .grad yet and without hooks.  Everything else reads it on the main stream
    at once: accumulation into an existing .grad, a non-leaf weight (the gradient travels on through view / permute nodes and
    is copied into the parameter's layout), tensor hooks, and DistributedDataParallel, whose per-parameter hook copies the
    gradient into its bucket as soon as it is accumulated (assumed whenever a process group is up)."""
    if not _DEFER_JOIN or not weight.is_leaf or weight.grad is not None or torch.cuda.is_current_stream_capturing():
        return False
    if getattr(weight, "_backward_hooks", None) or getattr(weight, "_post_accumulate_grad_hooks", None):
        return False
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        # pcdet_amd.utils.ddp_utils.wrap_ddp takes this weight out of DDP's buckets and reduces it after the end-of-pass join
        return bool(getattr(weight, "_spx_manual_reduce", False))
    return True


def _conv(src, wp, c_dst, kvol, pair, ld, n_dst, flip, scale, shift, relu, d_n, rb, want_stats=False):
    """One gather-GEMM launch: the ring schedule (round 3: every tile owned by one wave, csrc/conv_ring.hip) for the 64-channel
    layers, the MFMA-work-balanced persistent schedule where that one applies (128-channel BEV entry conv), the
    one-tile-per-wave kernel otherwise; plans are cached on the rulebook, over rows grouped by offset mask for submanifold
    tables (which several launches share).  want_stats: return (out, stats) where stats is the [rows, 2, c_dst] per-workgroup
    column sums of out and out**2 when the launch took them in its epilogue, else None."""
    out = stats = None
    if rb is not None and src.is_cuda and n_dst > 0 and src.shape[0] > 0:
        if ops.ring_ok(src.shape[1], c_dst, n_dst, kvol):
            perm, table, tld, plan = ops.ring_plan_for(rb, pair, ld, kvol, n_dst, d_n)
            out = ops.conv_gemm_ring(src, wp, c_dst, kvol, table, tld, n_dst, plan, flip_k=flip, scale=scale, shift=shift,
                                     relu=relu, d_n_dst=d_n, perm=perm, want_stats=want_stats)
            if want_stats:
                out, stats = out
        elif ops.balanced_ok(src.shape[1], c_dst, n_dst, rb, pair):
            if ops.grouped_ok(rb, kvol):
                perm, grouped, plan = ops.grouped_plan_for(rb, pair, ld, kvol, n_dst, d_n)
                out = ops.conv_gemm_balanced(src, wp, c_dst, kvol, grouped, n_dst, n_dst, plan, flip_k=flip, scale=scale,
                                             shift=shift, relu=relu, d_n_dst=d_n, perm=perm)
            else:
                plan = ops.plan_for(rb, pair, ld, kvol, n_dst, d_n)
                out = ops.conv_gemm_balanced(src, wp, c_dst, kvol, pair, ld, n_dst, plan, flip_k=flip, scale=scale,
                                             shift=shift, relu=relu, d_n_dst=d_n)
    if out is None:
        out = ops.conv_gemm(src, wp, c_dst, kvol, pair, ld, n_dst, flip_k=flip, scale=scale, shift=shift, relu=relu,
                            d_n_dst=d_n)
    return (out, stats) if want_stats else out


class _SparseConvFn(torch.autograd.Function):
    """out = gather(src, pair_f) (*) W   — one sparse conv application.

    pair_f/ld_f/n_dst : table used in the forward direction (rows of the result)
    pair_b/ld_b/flip_b: table used for dgrad (rows of the source); flip_b reads the forward table at K-1-k
    """

    @staticmethod
    def forward(ctx, feats, weight, bias, pair_f, ld_f, n_dst, pair_b, ld_b, flip_b, scale, shift, relu, d_n=None,
                rb=None, d_n_src=None):
        cout, cin = weight.shape[0], weight.shape[-1]
        kvol = weight.numel() // (cout * cin)
        wp = _packed(weight, 0)
        sh = shift if shift is not None else bias
        out = _conv(feats, wp, cout, kvol, pair_f, ld_f, n_dst, False, scale, sh, relu, d_n, rb)
        ctx.save_for_backward(feats, weight)
        ctx.rb, ctx.d_n_src, ctx.d_n = rb, d_n_src, d_n
        ctx.tables = (pair_f, ld_f, n_dst, pair_b, ld_b, flip_b)
        ctx.has_bias = bias is not None
        ctx.fused = scale is not None or shift is not None or relu
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.fused:
            raise RuntimeError("fused scale/shift/relu epilogue is inference-only")
        feats, weight = ctx.saved_tensors
        dfe, dw, db = _conv_backward(feats, weight, ctx.tables, ctx.rb, ctx.d_n_src, ctx.has_bias, dout,
                                     ctx.needs_input_grad[:3], ctx.d_n)
        return dfe, dw, db, None, None, None, None, None, None, None, None, None, None, None, None


def _conv_backward(feats, weight, tables, rb, d_n_src, has_bias, dout, needs, d_n_dst=None):
    """dgrad (the same gather-GEMM with transposed weights over the backward table), wgrad, bias gradient.  d_n_src /
    d_n_dst: device-side live row counts of the layer's input / output in static-capacity mode."""
    pair_f, ld_f, n_dst, pair_b, ld_b, flip_b = tables
    cout, cin = weight.shape[0], weight.shape[-1]
    kvol = weight.numel() // (cout * cin)
    dout = dout.contiguous()
    dfe = dw = db = None
    if needs[0]:
        wt = _packed(weight, 1)
        dfe = _conv(dout, wt, cin, kvol, pair_b, ld_b, feats.shape[0], flip_b, None, None, False, d_n_src, rb)
    if needs[1]:
        def _wgrad():
            counts = None
            if rb is not None and feats.is_cuda and n_dst > 0 and feats.shape[0] > 0:
                counts = ops.wgrad_counts_for(rb, pair_f, ld_f, kvol, n_dst, d_n_dst)
            return ops.conv_wgrad(feats, dout, pair_f, ld_f, n_dst, tuple(weight.shape), d_n_out=d_n_dst, counts=counts)
        if _ASYNC_WGRAD and feats.is_cuda and n_dst > 0 and feats.shape[0] > 0:
            dw = _off_critical_path(_wgrad, [feats, dout, pair_f], weight)     # enqueued after the data gradient above
        else:
            dw = _wgrad()
    if has_bias and needs[2]:
        if d_n_dst is not None:
            # rows beyond the live count are undefined (possibly NaN): select, do not multiply
            live = torch.arange(dout.shape[0], device=dout.device).unsqueeze(1) < d_n_dst
            db = torch.where(live, dout, torch.zeros((), dtype=dout.dtype, device=dout.device)).sum(0)
        else:
            db = dout.sum(0)
    return dfe, dw, db


_EPILOGUE_STATS = os.environ.get("SPX_CONV_EPILOGUE_STATS", "1") != "0"     # dev knob


class _SparseConvBNReLUFn(torch.autograd.Function):
    """Sparse conv + training-mode BatchNorm1d + ReLU of one `post_act_block` (reference spconv_backbone.py:9-35) as ONE
    autograd node: the same kernels as _SparseConvFn followed by _BNReLUFn, half the python / autograd bookkeeping per
    layer — the low-channel layers at the end of the backward pass are bound by that bookkeeping, not by their kernels
    (tools/host_overhead.py)."""

    @staticmethod
    def forward(ctx, feats, weight, bias, gamma, beta, running_mean, running_var, nbt, momentum, eps, relu, tables, rb,
                d_n, d_n_src):
        pair_f, ld_f, n_dst = tables[:3]
        cout, cin = weight.shape[0], weight.shape[-1]
        kvol = weight.numel() // (cout * cin)
        out, sums = _conv(feats, _packed(weight, 0), cout, kvol, pair_f, ld_f, n_dst, False, None, bias, False, d_n, rb,
                          want_stats=True)
        if sums is not None and _EPILOGUE_STATS:
            # the batch sums came out of the convolution's epilogue (ring schedule): no statistics pass over `out`
            y, mean, invstd = ops.bn_relu_fwd_from_sums(out, sums, gamma, beta, running_mean, running_var, momentum, eps, relu,
                                                        num_batches_tracked=nbt, d_n=d_n)
        else:
            y, mean, invstd = ops.bn_relu_fwd(out, gamma, beta, running_mean, running_var, momentum, eps, relu,
                                              num_batches_tracked=nbt, d_n=d_n)
        ctx.save_for_backward(feats, weight, out, gamma, beta, mean, invstd)
        ctx.tables, ctx.rb, ctx.d_n_src, ctx.has_bias, ctx.relu = tables, rb, d_n_src, bias is not None, relu
        ctx.d_n = d_n
        ctx.mark_non_differentiable(running_mean, running_var)
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, weight, out, gamma, beta, mean, invstd = ctx.saved_tensors
        dout, dgamma, dbeta = ops.bn_relu_bwd(out, dy, gamma, beta, mean, invstd, ctx.relu, d_n=ctx.d_n)
        dfe, dw, db = _conv_backward(feats, weight, ctx.tables, ctx.rb, ctx.d_n_src, ctx.has_bias, dout,
                                     ctx.needs_input_grad[:3], ctx.d_n)
        return dfe, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def sparse_conv_bn_relu(feats, weight, bias, rb, bn, relu):
    """sparse_conv (not inverse) followed by `bn` in training mode and optionally ReLU, as one autograd node."""
    tables = ((rb.pair, rb.ld, rb.n_out, rb.pair, rb.ld, True) if rb.subm else
              (rb.pair, rb.ld, rb.n_out, rb.pair_bwd, rb.pair_bwd.shape[1], False))
    mom = bn.momentum          # never None here: bn_momentum_ok() gates every fused path
    return _SparseConvBNReLUFn.apply(feats, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     bn.num_batches_tracked, mom, bn.eps, relu, tables, rb, rb.d_n_out, rb.d_n_in)


def sparse_conv(feats, weight, bias, rb, inverse=False, scale=None, shift=None, relu=False):
    """Apply rulebook `rb` (spx.ops.Rulebook).  inverse=True swaps the roles of the two tables
    (SparseInverseConv3d: outputs are the strided conv's inputs)."""
    if not inverse:
        if rb.subm:
            return _SparseConvFn.apply(feats, weight, bias, rb.pair, rb.ld, rb.n_out, rb.pair, rb.ld, True, scale,
                                       shift, relu, rb.d_n_out, rb, rb.d_n_in)
        return _SparseConvFn.apply(feats, weight, bias, rb.pair, rb.ld, rb.n_out, rb.pair_bwd, rb.pair_bwd.shape[1],
                                   False, scale, shift, relu, rb.d_n_out, rb, rb.d_n_in)
    assert not rb.subm and rb.pair_bwd is not None
    return _SparseConvFn.apply(feats, weight, bias, rb.pair_bwd, rb.pair_bwd.shape[1], rb.n_in, rb.pair, rb.ld, False,
                               scale, shift, relu, rb.d_n_in, rb, rb.d_n_out)


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, indices, batch_size, spatial_shape, channels_last, d_n=None):
        ctx.args = (indices, batch_size, list(spatial_shape), channels_last, d_n)
        return ops.densify(feats, indices, batch_size, spatial_shape, channels_last, d_n=d_n)

    @staticmethod
    def backward(ctx, ddense):
        indices, batch_size, spatial_shape, channels_last, d_n = ctx.args
        return (ops.densify_bwd(ddense, indices, batch_size, spatial_shape, channels_last, d_n=d_n), None, None, None, None,
                None)


def dense(feats, indices, batch_size, spatial_shape, channels_last=False, d_n=None):
    return _DenseFn.apply(feats, indices, batch_size, spatial_shape, channels_last, d_n)


class _BNReLUFn(torch.autograd.Function):
    """Training-mode BatchNorm1d (+ residual) (+ReLU) on sparse feature rows: libspx kernels forward and backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual=None, nbt=None, d_n=None):
        y, mean, invstd = ops.bn_relu_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual,
                                          num_batches_tracked=nbt, d_n=d_n)
        ctx.d_n = d_n
        if residual is None:
            ctx.save_for_backward(x, gamma, beta, mean, invstd)
        else:
            ctx.save_for_backward(x, gamma, beta, mean, invstd, residual)
        ctx.relu = relu
        ctx.mark_non_differentiable(running_mean, running_var) if running_mean is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors[:5]
        residual = ctx.saved_tensors[5] if len(ctx.saved_tensors) > 5 else None
        out = ops.bn_relu_bwd(x, dy, gamma, beta, mean, invstd, ctx.relu, residual, d_n=ctx.d_n)
        return (out[0], out[1], out[2], None, None, None, None, None, (out[3] if residual is not None else None), None,
                None)


class _BNReLUCatFn(torch.autograd.Function):
    """relu(bn_i(x_i)) for several row matrices x_i [N, C_i] written side by side into ONE [N, sum C_i] matrix: the
    channel concatenation of the BEV up-sampling branches (reference base_bev_backbone.py:99-106: deblocks, then
    torch.cat(ups, dim=1)) without the concatenation copy, and backward without the slice copies (each layer reads its
    gradient from its channel slice in place)."""

    @staticmethod
    def forward(ctx, momenta, epss, *args):
        k = len(momenta)
        xs, gammas, betas = args[:k], args[k:2 * k], args[2 * k:3 * k]
        rms, rvs, nbts = args[3 * k:4 * k], args[4 * k:5 * k], args[5 * k:6 * k]
        n = xs[0].shape[0]
        widths = [x.shape[1] for x in xs]
        out = torch.empty((n, sum(widths)), dtype=torch.float32, device=xs[0].device)
        saved, off = [], 0
        for i in range(k):
            _y, mean, invstd = ops.bn_relu_fwd(xs[i], gammas[i], betas[i], rms[i], rvs[i], momenta[i], epss[i], True,
                                               out=out[:, off:off + widths[i]], num_batches_tracked=nbts[i])
            saved += [xs[i].contiguous(), gammas[i], betas[i], mean, invstd]
            off += widths[i]
        ctx.save_for_backward(*saved)
        ctx.widths = widths
        ctx.mark_non_differentiable(*[t for t in rms + rvs + nbts if t is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        k = len(ctx.widths)
        s = ctx.saved_tensors
        if dout.stride(1) != 1 or dout.stride(0) % 4 != 0 or dout.data_ptr() % 16 != 0:
            dout = dout.contiguous()
        dxs, dgs, dbs, off = [], [], [], 0
        for i in range(k):
            x, gamma, beta, mean, invstd = s[5 * i:5 * i + 5]
            dx, dg, db = ops.bn_relu_bwd(x, dout[:, off:off + ctx.widths[i]], gamma, beta, mean, invstd, True)
            dxs.append(dx), dgs.append(dg), dbs.append(db)
            off += ctx.widths[i]
        return (None, None) + tuple(dxs) + tuple(dgs) + tuple(dbs) + (None,) * (3 * k)


def bn_relu_cat_train(xs, bns):
    """[relu(bn_i(x_i))] concatenated along the channel axis -> [N, sum C_i]; bns: nn.BatchNorm modules in training mode
    that libspx's kernels cover (see bn_train_fusable for the conditions on each pair)."""
    moms = tuple(bn.momentum for bn in bns)                         # callers check bn_train_fusable: never None
    epss = tuple(bn.eps for bn in bns)
    args = (tuple(xs) + tuple(bn.weight for bn in bns) + tuple(bn.bias for bn in bns) + tuple(bn.running_mean for bn in bns)
            + tuple(bn.running_var for bn in bns) + tuple(bn.num_batches_tracked for bn in bns))
    return _BNReLUCatFn.apply(moms, epss, *args)


def bn_momentum_ok(bn):
    """momentum=None means a CUMULATIVE moving average in torch (factor 1 / num_batches_tracked): the fused kernels take a
    fixed factor, so such a module stays on the torch path."""
    return bn.momentum is not None


def bn_train_fusable(bn, f):
    """True when libspx's training-mode BatchNorm kernels cover `bn` applied to the rows of f [N, C]."""
    return (isinstance(bn, torch.nn.BatchNorm1d) and bn.training and bn.affine and bn.track_running_stats
            and bn_momentum_ok(bn) and f.is_cuda
            and f.dtype == torch.float32 and f.dim() == 2 and f.shape[0] > 1 and ops.bn_relu_supported(f.shape[1]))


def bn_relu_train(x, bn, relu, residual=None, d_n=None):
    """x [N, C] through `bn` (nn.BatchNorm1d in training mode, affine, tracking running stats), plus `residual` when
    given, and optionally ReLU.  d_n: device-side live row count of a static-capacity tensor."""
    # num_batches_tracked is incremented inside the finalize kernel (one launch less per layer)
    return _BNReLUFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu, residual,
                           bn.num_batches_tracked, d_n)


def bn_act(x, bn, relu, residual=None, d_n=None):
    """relu(bn(x) + residual) on feature rows: the fused kernels when they apply (training on the GPU), else the torch
    modules in the reference's order (spconv_backbone.py:56-72).  d_n: live row count of a static-capacity tensor (only
    the fused kernels can honour it)."""
    if bn_train_fusable(bn, x):
        return bn_relu_train(x, bn, relu, residual, d_n)
    if d_n is not None:
        if bn.training or bn.running_mean is None:
            raise RuntimeError("static-capacity rows need the fused BatchNorm kernels (training mode on the GPU, fp32) or a "
                               "BatchNorm on its running statistics")
        # BatchNorm on its running statistics while the model trains (frozen BatchNorm): row-local torch modules with the rows
        # beyond the live count (garbage, possibly NaN, and so is their gradient) replaced by zeros before and after — selects,
        # not multiplies — so that neither the output nor the parameter gradients (sums over rows) see them
        live = (torch.arange(x.shape[0], device=x.device) < d_n).unsqueeze(1)
        zero = torch.zeros((), dtype=x.dtype, device=x.device)
        y = bn(torch.where(live, x, zero))
        if residual is not None:
            y = y + torch.where(live, residual, zero)
        y = torch.relu(y) if relu else y
        return torch.where(live, y, zero)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y

# ---------------------------------------------------------------------------------------------------------------------
# dense 3x3 / stride 1 / pad 1 convolution of the BEV backbone on libspx's Winograd kernel (csrc/wino_conv2d.hip)

def _wino_image(weight, flip):
    """Transformed weight image of `weight` (flip: of its data-gradient filter), in a buffer that lives ON the weight tensor
    (attribute) and is re-filled when the weight has changed: once per call in training (the optimizer steps in between;
    ~10 us), once in all at inference.  One persistent buffer per layer and direction — no allocation per call."""
    # Capture FIRST, as _packed does: a graph must record the transform kernel into a buffer of its own.  (Round 2 looked at
    # the cache first: GraphedTrainStep warms up without an optimizer step, so the forward of every BEV conv hit the cache,
    # no spx_wino_weight was recorded, and every replay after the first optimizer.step() convolved with the weights of capture
    # time while dgrad / wgrad used the current ones.)
    if weight.is_cuda and torch.cuda.is_current_stream_capturing():
        hit = getattr(weight, '_spx_wino_flip' if flip else '_spx_wino', None)
        if (not torch.is_grad_enabled() and hit is not None and hit[0] == weight._version and hit[1] == weight.data_ptr()
                and hit[2].device == weight.device):
            # an inference graph reads the image the warm-up passes cached; refresh_graph_constants() keeps it current
            return _graph_constant(weight, hit[2], lambda w, out, flip=flip: ops.wino_weight(w, flip, out=out))
        return ops.wino_weight(weight, flip)            # a graph that trains owns its buffers
    attr = '_spx_wino_flip' if flip else '_spx_wino'
    hit = getattr(weight, attr, None)
    fresh = hit is not None and hit[1] == weight.data_ptr() and hit[2].device == weight.device
    if fresh and hit[0] == weight._version and not (torch.is_grad_enabled() and weight.requires_grad):
        return hit[2]
    u = ops.wino_weight(weight, flip, out=hit[2] if fresh else None)
    setattr(weight, attr, (weight._version, weight.data_ptr(), u))
    return u


def _wino_image_always(weight, flip):
    """Backward runs with grad mode off; the image must still follow the weight of THIS step."""
    with torch.enable_grad():
        return _wino_image(weight, flip)


def wino_conv2d_ok(x, conv):
    """nn.Conv2d `conv` over `x` is one spx_conv2d_wino call: 3x3, stride 1, padding 1, no bias / groups / dilation, fp32
    channels-last map on the GPU, channel counts the kernel takes."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_autocast_enabled()
            and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and conv.padding == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is None and conv.padding_mode == 'zeros'
            and conv.weight.dtype == torch.float32 and ops.wino_ok(conv.in_channels, conv.out_channels)
            and x.shape[1] == conv.in_channels and ops._cl_ld(x) is not None)


_WINO_WGRAD = os.environ.get("SPX_BEV_WINOGRAD_WGRAD", "1") != "0"      # dev knob
class _WinoConv2dFn(torch.autograd.Function):
    """y = conv2d(x, w, padding=1): forward, data gradient (the same kernel over dy with the rotated / transposed filter
    image) and weight gradient (csrc/wino_wgrad.hip) in the Winograd domain; shapes the weight-gradient kernel does not
    take go to the vendor library (aten.convolution_backward with only the weight mask set)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return ops.conv2d_wino(x, _wino_image(weight, False), weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        return _wino_backward(x, weight, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


def _wino_backward(x, weight, dy, needs_x, needs_w):
    """Data gradient and weight gradient of conv2d(x, weight, padding=1) given dy, each on the Winograd kernels when its shape
    condition holds, else through the vendor library."""
    dx = dw = None
    if ops._cl_ld(dy) is None:
        dy = dy.contiguous(memory_format=torch.channels_last)
    if needs_x:
        if ops.wino_ok(weight.shape[0], weight.shape[1]):
            dx = ops.conv2d_wino(dy, _wino_image_always(weight, True), weight.shape[1])
        else:
            dx = torch.ops.aten.convolution_backward(dy, x, weight, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                     (True, False, False))[0]
    if needs_w:
        if _WINO_WGRAD and ops.wino_wgrad_ok(weight.shape[1], weight.shape[0], x.shape[3]):
            if _ASYNC_WGRAD:   # enqueued after the data gradient: that one gets the CUs first, this one what it frees
                dw = _off_critical_path(lambda: ops.conv2d_wino_wgrad(x, dy, weight), [x, dy], weight)
            else:
                dw = ops.conv2d_wino_wgrad(x, dy, weight)
        else:
            dw = torch.ops.aten.convolution_backward(dy, x, weight, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                     (False, True, False))[1]
    return dx, dw


class _WinoConvBNReLUFn(torch.autograd.Function):
    """Conv2d(3x3, padding 1) + training-mode BatchNorm2d + ReLU of one BEV-backbone layer (reference
    base_bev_backbone.py:45-49) as ONE autograd node.  The convolution kernel's epilogue takes the per-channel sums of y and
    y*y while it stores y, so BatchNorm needs no statistics pass over the map (one full read of it less per layer)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, nbt, momentum, eps):
        y, part = ops.conv2d_wino(x, _wino_image(weight, False), weight.shape[0], stats=True)
        b, c, h, w = y.shape
        rows = y.permute(0, 2, 3, 1).reshape(b * h * w, c)
        act, mean, invstd = ops.bn_relu_fwd_from_sums(rows, part, gamma, beta, running_mean, running_var, momentum, eps,
                                                      True, num_batches_tracked=nbt)
        ctx.save_for_backward(x, weight, rows, gamma, beta, mean, invstd)
        ctx.shape = (b, c, h, w)
        ctx.mark_non_differentiable(running_mean, running_var)
        return act.view(b, h, w, c).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, da):
        x, weight, rows, gamma, beta, mean, invstd = ctx.saved_tensors
        b, c, h, w = ctx.shape
        da_rows = da.permute(0, 2, 3, 1).reshape(b * h * w, c)
        drows, dgamma, dbeta = ops.bn_relu_bwd(rows, da_rows, gamma, beta, mean, invstd, True)
        dy = drows.view(b, h, w, c).permute(0, 3, 1, 2)
        dx, dw = _wino_backward(x, weight, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw, dgamma, dbeta, None, None, None, None, None


def wino_conv_bn_relu_train(x, conv, bn):
    """relu(bn(conv(x))) for a Winograd-eligible Conv2d and a BatchNorm2d in training mode (affine, running statistics, fixed
    momentum) as one autograd node."""
    return _WinoConvBNReLUFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, bn.momentum, bn.eps)


def wino_conv2d(x, weight, scale=None, shift=None, relu=False):
    """conv2d(x, weight, padding=1) on the Winograd kernel; scale / shift / relu: the folded inference epilogue
    relu?(y * scale[c] + shift[c]) (no autograd through it)."""
    if scale is not None or shift is not None or relu:
        assert not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad))
        return ops.conv2d_wino(x, _wino_image(weight, False), weight.shape[0], scale=scale, shift=shift, relu=relu)
    return _WinoConv2dFn.apply(x, weight)
