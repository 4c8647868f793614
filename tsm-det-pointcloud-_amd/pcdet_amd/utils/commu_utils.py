"""Cross-rank helpers of the data-parallel training loop (subset of the reference's pcdet/utils/commu_utils.py that
tools/train_utils/train_utils.py:82-84 and the eval loop use): world size / rank, barrier, gather of small python
values, scalar average.  One process per GPU; the process group is whatever init_dist_pytorch created (backend "nccl" =
RCCL on ROCm, "gloo" in the CPU tests).  Collectives of python objects go through all_gather_object, so they work on
either backend without staging through `.cuda()` as the reference does."""
import torch
import torch.distributed as dist


def _active():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _active() else 1


def get_rank():
    return dist.get_rank() if _active() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if _active() and dist.get_world_size() > 1:
        dist.barrier()


def all_gather(data):
    """list with `data` of every rank (any picklable value; tensors come back on the CPU)."""
    world = get_world_size()
    if world == 1:
        return [data]
    if isinstance(data, torch.Tensor):
        data = data.detach().cpu()
    out = [None] * world
    dist.all_gather_object(out, data)
    return out


def average_reduce_value(data):
    """mean over ranks of a python scalar (the timing meters of train_one_epoch, train_utils.py:82-84)."""
    vals = all_gather(data)
    return sum(vals) / len(vals)


def all_reduce(data, op="sum", average=False):
    """All-reduce of a tensor over the ranks; op in sum / max / min / product.  As the reference's helper
    (pcdet/utils/commu_utils.py:148-168) the argument is NOT modified: the reduction runs on a clone and, with average=True,
    a new tensor reduced / world_size is returned (so integer tensors average to floats instead of raising)."""
    world = get_world_size()
    if world == 1:
        return data
    ops = {"SUM": dist.ReduceOp.SUM, "MAX": dist.ReduceOp.MAX, "MIN": dist.ReduceOp.MIN, "PRODUCT": dist.ReduceOp.PRODUCT}
    reduced = data.clone()
    dist.all_reduce(reduced, op=ops[op.upper()])
    if average:
        assert op.upper() == "SUM"
        return reduced / world
    return reduced
