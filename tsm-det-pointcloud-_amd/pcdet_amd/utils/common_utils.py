"""Small helpers the hot path needs (subset of the reference's pcdet/utils/common_utils.py)."""
import logging
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def limit_period(val, offset=0.5, period=np.pi):
    """val - floor(val/period + offset) * period  (reference common_utils.py:27-30)."""
    is_np = isinstance(val, np.ndarray)
    t = torch.from_numpy(val).float() if is_np else val
    out = t - torch.floor(t / period + offset) * period
    return out.numpy() if is_np else out


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """(z,y,x) voxel indices -> metric xyz centres (reference common_utils.py:73-89)."""
    assert voxel_coords.shape[1] == 3
    xyz = voxel_coords[:, [2, 1, 0]].float()
    vs = torch.tensor(voxel_size, device=xyz.device).float() * downsample_times
    lo = torch.tensor(point_cloud_range[0:3], device=xyz.device).float()
    return (xyz + 0.5) * vs + lo


def mask_points_by_range(points, limit_range):
    """reference common_utils.py:66-70 (upper bound hi - 0.0002)."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= (limit_range[3] - 0.0002)) \
        & (points[:, 1] >= limit_range[1]) & (points[:, 1] <= (limit_range[4] - 0.0002)) \
        & (points[:, 2] >= limit_range[2]) & (points[:, 2] <= (limit_range[5] - 0.0002))


def create_logger(log_file=None, rank=0, log_level=logging.INFO):
    logger = logging.getLogger("pcdet_amd")
    logger.setLevel(log_level if rank == 0 else "ERROR")
    if not logger.handlers:
        fmt = logging.Formatter("%(asctime)s  %(levelname)5s  %(message)s")
        console = logging.StreamHandler()
        console.setFormatter(fmt)
        logger.addHandler(console)
        if log_file is not None:
            fh = logging.FileHandler(filename=log_file)
            fh.setFormatter(fmt)
            logger.addHandler(fh)
    logger.propagate = False
    return logger


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class AverageMeter(object):
    """running value / mean of a scalar (the d_time / f_time / b_time meters of the reference's train_one_epoch,
    tools/train_utils/train_utils.py:19-22; reference common_utils.py:283-299)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist_pytorch(tcp_port=None, local_rank=None, backend="nccl"):
    """One process per GPU; backend 'nccl' IS RCCL on PyTorch-ROCm (reference common_utils.py:184-199).
    Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun)."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if backend == "nccl":
        num_gpus = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % max(num_gpus, 1))
    if tcp_port is not None and "MASTER_PORT" not in os.environ:
        os.environ["MASTER_PORT"] = str(tcp_port)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


def scatter_point_inds(indices, point_inds, shape):
    """Dense int32 table of `shape`, -1 everywhere except table[indices] = point_inds (reference spconv_utils.py)."""
    table = torch.full(tuple(int(x) for x in shape), -1, dtype=point_inds.dtype, device=point_inds.device)
    table[tuple(indices[:, j] for j in range(indices.shape[1]))] = point_inds
    return table


def generate_voxel2pinds(sparse_tensor):
    """[B, Z, Y, X] table: row of the active voxel at each cell, -1 elsewhere (reference common_utils.py:257-265); what
    voxel_query and the voxel-centroid lookup index into."""
    idx = sparse_tensor.indices.long()
    rows = torch.arange(idx.shape[0], device=idx.device, dtype=torch.int32)
    return scatter_point_inds(idx, rows, [sparse_tensor.batch_size] + list(sparse_tensor.spatial_shape))
