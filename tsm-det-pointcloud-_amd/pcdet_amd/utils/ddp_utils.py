"""DistributedDataParallel with the convolutions' weight gradients kept off the critical path of the backward pass.

Reference: tools/train.py:154-155 wraps the detector in `nn.parallel.DistributedDataParallel(model, device_ids=[...])`.  That
still works here unchanged.  What it costs on this library: DDP's per-parameter hook copies a gradient into its bucket on the
compute stream as soon as autograd has accumulated it, so the weight-gradient kernels — which libspx launches on a stream of
their own and joins at the END of the backward pass (spx/functional.py:_off_critical_path) — have to be joined right after every
launch instead: the backward pass runs serialised, 284 instead of 305 frames/s per GPU (measured on one MI355X,
SPX_WGRAD_DEFER_JOIN=0).

`wrap_ddp(model, **kw)` keeps the overlap: the weights of the sparse and 3x3 dense convolutions are excluded from DDP's buckets
(`_set_params_and_buffers_to_ignore_for_model`, the supported way to hand a parameter's reduction to the caller), keep their
deferred join, and are all-reduced as ONE flat buffer at the end of the backward pass, after the join (autograd engine
callback, armed by DDP's communication hook so that it follows DDP's own `no_sync()` / accumulation semantics).  Everything else
— BatchNorm parameters, the head, buffers — stays with DDP.  The result in every `.grad` is the same average over ranks.
"""
import torch
import torch.distributed as dist
from torch import nn
from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
from torch.nn.parallel import DistributedDataParallel


def late_reduced_parameters(model):
    """{name: parameter} of the weights whose gradient kernels libspx runs on its side stream: SparseConvolution weights and the
    3x3 Conv2d weights of the BEV backbone (Winograd weight gradient; the first conv of a block runs as a sparse conv).  The
    list does not have to be exact: a listed weight whose gradient autograd computes in stream order is reduced just the same."""
    from spx.modules import is_sparse_conv
    out = {}
    for mname, m in model.named_modules():
        w = getattr(m, "weight", None)
        if not isinstance(w, nn.Parameter) or not w.requires_grad:
            continue
        if is_sparse_conv(m) or (isinstance(m, nn.Conv2d) and tuple(m.kernel_size) == (3, 3) and m.groups == 1):
            out[(mname + "." if mname else "") + "weight"] = w
    return out


class _LateReducer(object):
    def __init__(self, ddp, params, group):
        self.params = list(params)
        self.group = group
        self.world = dist.get_world_size(group)
        self.armed = False
        ddp.register_comm_hook(self, _LateReducer._hook)

    @staticmethod
    def _hook(self, bucket):
        # DDP calls this for every bucket it reduces, from inside the backward pass — exactly when gradients are being
        # synchronised (not under no_sync()): arm the end-of-pass reduction once, then do what DDP does by default
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.reduce)
        return default_hooks.allreduce_hook(self.group, bucket)

    def reduce(self):
        self.armed = False
        ps = [p for p in self.params if p.grad is not None]
        if not ps:
            return
        dev = ps[0].grad.device
        if dev.type == "cuda":
            from spx import functional as F_
            side = F_._SIDE_STREAMS.get(dev.index)
            if side is not None:                       # the kernels that produce these gradients (idempotent with the
                torch.cuda.current_stream(dev).wait_stream(side)     # deferred joins' own callbacks)
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, group=self.group)
        flat.div_(self.world)
        # back into the gradients' own tensors (they keep the parameters' layout — channels_last Conv2d weights —, which the
        # fused optimizers insist on)
        views, off = [], 0
        for p in ps:
            n = p.numel()
            views.append(flat[off:off + n].view(p.grad.shape))
            off += n
        torch._foreach_copy_([p.grad for p in ps], views)


def wrap_ddp(model, process_group=None, late_reduce=True, **ddp_kwargs):
    """DistributedDataParallel(model, **ddp_kwargs) with the convolution weights reduced at the end of the backward pass (see the
    module docstring).  late_reduce=False: plain DDP."""
    params = late_reduced_parameters(model) if late_reduce else {}
    late_ids = {id(p) for p in params.values()}
    if not any(p.requires_grad and id(p) not in late_ids for p in model.parameters()):
        params = {}        # nothing would be left for DDP itself (its communication hook arms the late reduction): plain DDP
    if params:
        DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(model, list(params))
        with torch.no_grad():                          # DDP does not broadcast what it ignores: rank 0's values everywhere
            flat = torch.cat([p.reshape(-1) for p in params.values()])
            dist.broadcast(flat, 0, group=process_group)
            off = 0
            for p in params.values():
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
    ddp = DistributedDataParallel(model, process_group=process_group, **ddp_kwargs)
    if params:
        for p in params.values():
            p._spx_manual_reduce = True                # spx.functional: nobody reads this gradient before the pass ends
        ddp._spx_late_reducer = _LateReducer(ddp, params.values(), process_group if process_group is not None else dist.group.WORLD)
    return ddp
