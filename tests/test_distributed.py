"""N > 1 path on CPU: world_size-2 gloo, one process per rank, frames sharded across ranks, the only collective is
DDP's gradient all-reduce.  The sparse ops run through the oracle backend here (no GPU in this
container); on the GPU box the same code path runs libspx under backend 'nccl' (= RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle.cpu_backend import use_oracle_backend
    from pcdet_amd.config import AttrDict, cfg_from_yaml_file
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models import build_network
    from pcdet_amd.utils import common_utils
    common_utils.init_dist_pytorch(backend="gloo")
    assert common_utils.get_dist_info() == (rank, world)
    cfg = cfg_from_yaml_file(os.path.join(ROOT, "tsm-det-pointcloud-_amd/tools/cfgs/kitti_models/second.yaml"), AttrDict())
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, True, cfg_id=0, length=4)
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, 3, ds)
    model.train()
    ddp = torch.nn.parallel.DistributedDataParallel(model, bucket_cap_mb=32, gradient_as_bucket_view=True)
    # DistributedSampler-style sharding: rank r takes frames r, r + world, ...
    frames = [ds[i] for i in range(rank, 4, world)][:1]
    b = ds.collate_batch(frames)
    bd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    with use_oracle_backend():
        ret, _tb, _ = ddp(bd)
        ret["loss"].backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    torch.save({"loss": ret["loss"].detach(), "grads": grads}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _single(frame_ids):
    from oracle.cpu_backend import use_oracle_backend
    from pcdet_amd.config import AttrDict, cfg_from_yaml_file
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models import build_network
    cfg = cfg_from_yaml_file(os.path.join(ROOT, "tsm-det-pointcloud-_amd/tools/cfgs/kitti_models/second.yaml"), AttrDict())
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, True, cfg_id=0, length=4)
    out = []
    # same intra-op thread count as the workers: with train-mode BatchNorm, a different fp32 summation order alone
    # moves some gradients by ~3e-3 (measured), which would mask what this test checks
    torch.set_num_threads(2)
    for fid in frame_ids:
        torch.manual_seed(0)
        model = build_network(cfg.MODEL, 3, ds)
        model.train()
        b = ds.collate_batch([ds[fid]])
        bd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}
        with use_oracle_backend():
            ret, _tb, _ = model(bd)
            ret["loss"].backward()
        out.append({n: p.grad.clone() for n, p in model.named_parameters()})
    return out


def test_ddp_gloo_world2_gradient_allreduce(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=True)
    # after the all-reduce every rank holds the same (averaged) gradients
    for n in r0["grads"]:
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
    # ... and they are the mean of the per-rank single-process gradients (frames 0 and 1)
    g0, g1 = _single([0, 1])
    worst = 0.0
    for n in g0:
        want = 0.5 * (g0[n] + g1[n])
        worst = max(worst, float((r0["grads"][n] - want).abs().max() / want.abs().max().clamp_min(1e-12)))
    assert worst < 1e-4, worst
    assert float(r0["loss"]) != float(r1["loss"])   # different frames per rank (weak scaling)


# ---------------------------------------------------------------------------------------------------------------------------
# pcdet_amd.utils.ddp_utils.wrap_ddp: convolution weights outside DDP's buckets, reduced as one flat buffer at the end of the pass

def _tiny_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, padding=1, bias=False), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.Flatten(), torch.nn.Linear(8 * 6 * 6, 3))


def _tiny_data(i):
    g = torch.Generator().manual_seed(100 + i)
    return torch.randn(2, 4, 6, 6, generator=g), torch.randn(2, 3, generator=g)


def _late_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from pcdet_amd.utils.ddp_utils import late_reduced_parameters, wrap_ddp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _tiny_model()
    late = late_reduced_parameters(model)
    assert sorted(late) == ["0.weight", "3.weight"]
    if rank == 1:
        with torch.no_grad():
            model[0].weight.add_(1.0)                   # must be overwritten by rank 0's values
    ddp = wrap_ddp(model)
    assert all(getattr(p, "_spx_manual_reduce", False) for p in late.values())
    out = {}
    # one synchronised step; then two accumulated micro-steps under no_sync() + a synchronised one
    x, y = _tiny_data(rank)
    torch.nn.functional.mse_loss(ddp(x), y).backward()
    out["step"] = {n: p.grad.clone() for n, p in model.named_parameters()}
    out["w0"] = model[0].weight.detach().clone()
    ddp.zero_grad(set_to_none=True)
    with ddp.no_sync():
        x, y = _tiny_data(10 + rank)
        torch.nn.functional.mse_loss(ddp(x), y).backward()
    x, y = _tiny_data(20 + rank)
    torch.nn.functional.mse_loss(ddp(x), y).backward()
    out["accum"] = {n: p.grad.clone() for n, p in model.named_parameters()}
    torch.save(out, os.path.join(out_dir, "late%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_wrap_ddp_late_reduction_world2(tmp_path):
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_late_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "late0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(tmp_path, "late1.pt"), weights_only=True)
    assert torch.equal(r0["w0"], r1["w0"]) and torch.equal(r0["w0"], _tiny_model()[0].weight.detach())
    torch.set_num_threads(1)

    def grads(ids):
        m = _tiny_model()
        for i in ids:
            x, y = _tiny_data(i)
            torch.nn.functional.mse_loss(m(x), y).backward()
        return {n: p.grad.clone() for n, p in m.named_parameters()}
    for key, per_rank in (("step", ([0], [1])), ("accum", ([10, 20], [11, 21]))):
        a, b = grads(per_rank[0]), grads(per_rank[1])
        for n in a:
            assert torch.equal(r0[key][n], r1[key][n]), (key, n)              # every rank holds the same gradients
            want = 0.5 * (a[n] + b[n])                                        # ... the mean over ranks, late-reduced or not
            assert float((r0[key][n] - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), (key, n)
