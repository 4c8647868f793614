"""Host-side logic (no GPU): config system, synthetic dataset contract, one-cycle schedule, module registries."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "tools", "cfgs")


def test_config_yaml_base_merge_and_overrides():
    from pcdet_amd.config import AttrDict, cfg_from_list, cfg_from_yaml_file
    cfg = cfg_from_yaml_file(os.path.join(CFG, "kitti_models", "second.yaml"), AttrDict())
    assert cfg.MODEL.NAME == "SECONDNet" and cfg.MODEL.BACKBONE_3D.NAME == "VoxelBackBone8x"
    assert cfg.DATA_CONFIG.POINT_CLOUD_RANGE == [0, -40, -3, 70.4, 40, 1]           # from _BASE_CONFIG_
    assert cfg.DATA_CONFIG.DATA_PROCESSOR[2].VOXEL_SIZE == [0.05, 0.05, 0.1]          # dicts inside lists are wrapped
    assert cfg.MODEL.get("ROI_HEAD", None) is None
    cfg_from_list(["OPTIMIZATION.LR", "0.01", "MODEL.BACKBONE_2D.LAYER_NUMS", "3,3",
                   "MODEL.DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS", "cls_weight:2.0"], cfg)
    assert cfg.OPTIMIZATION.LR == 0.01 and cfg.MODEL.BACKBONE_2D.LAYER_NUMS == [3, 3]
    assert cfg.MODEL.DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS.cls_weight == 2.0
    with pytest.raises(AssertionError):
        cfg_from_list(["NO.SUCH.KEY", "1"], cfg)
    w = cfg_from_yaml_file(os.path.join(CFG, "waymo_models", "second.yaml"), AttrDict())
    assert w.CLASS_NAMES[0] == "Vehicle" and w.DATA_CONFIG.POINT_CLOUD_RANGE[0] == -75.2


def test_synthetic_generator_is_deterministic_and_exact():
    from pcdet_amd.datasets import SyntheticDataset, synthetic
    a, b = synthetic.make_frame(2, 5), synthetic.make_frame(2, 5)
    assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["gt_boxes"], b["gt_boxes"])
    assert not np.array_equal(a["points"], synthetic.make_frame(2, 6)["points"])
    assert a["n_active"] == 16000 and a["points"].shape[1] == 4 and a["gt_boxes"].shape == (10, 8)
    g = synthetic.KITTI
    lo, hi = np.array(g["point_cloud_range"][:3]), np.array(g["point_cloud_range"][3:])
    assert (a["points"][:, :3] >= lo).all() and (a["points"][:, :3] < hi).all()
    ds = SyntheticDataset(cfg_id=0)
    batch = ds.collate_batch([ds[0], ds[1], ds[2]])
    assert batch["batch_size"] == 3 and batch["points"].shape[1] == 5
    assert np.array_equal(np.unique(batch["points"][:, 0]), [0.0, 1.0, 2.0])
    assert (np.diff(batch["points"][:, 0]) >= 0).all()                       # frames contiguous and ascending
    assert list(ds.grid_size) == [256, 320, 40]


def test_one_cycle_schedule_shape():
    from pcdet_amd.config import AttrDict
    from tools.train_utils.optimization import build_optimizer, build_scheduler
    m = torch.nn.Linear(4, 4)
    oc = AttrDict(OPTIMIZER="adam_onecycle", LR=0.003, WEIGHT_DECAY=0.01, MOMS=[0.95, 0.85], PCT_START=0.4,
                  DIV_FACTOR=10, DECAY_STEP_LIST=[35, 45], LR_DECAY=0.1, LR_CLIP=1e-7, MOMENTUM=0.9)
    opt = build_optimizer(m, oc)
    sched, _ = build_scheduler(opt, 100, 1, -1, oc)
    lrs = [sched.step(i) for i in range(101)]
    assert abs(lrs[0] - 0.0003) < 1e-9 and abs(max(lrs) - 0.003) < 1e-9 and int(np.argmax(lrs)) == 40
    assert lrs[100] < 1e-6 and all(x <= y + 1e-12 for x, y in zip(lrs[:40], lrs[1:41]))
    sched.step(40)
    assert abs(opt.param_groups[0]["betas"][0] - 0.85) < 1e-9
    sched.step(0)
    assert abs(opt.param_groups[0]["betas"][0] - 0.95) < 1e-9


def test_registries_and_unknown_modules():
    from pcdet_amd.config import AttrDict, cfg_from_yaml_file
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models import backbones_3d, build_network
    assert set(backbones_3d.__all__) == {"VoxelBackBone8x", "VoxelResBackBone8x", "UNetV2"}
    cfg = cfg_from_yaml_file(os.path.join(CFG, "kitti_models", "second.yaml"), AttrDict())
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, True, cfg_id=0)
    cfg.MODEL.BACKBONE_3D.NAME = "VoxelResBackBone8x"
    model = build_network(cfg.MODEL, 3, ds)
    assert model.backbone_3d.backbone_channels["x_conv4"] == 128
    cfg.MODEL.ROI_HEAD = AttrDict(NAME="PVRCNNHead", CLASS_AGNOSTIC=True)
    with pytest.raises(NotImplementedError):
        build_network(cfg.MODEL, 3, ds)                                       # second-stage heads are out of scope: loud


def test_sparse_tensor_container_api():
    import spx
    t = spx.SparseConvTensor(torch.zeros(3, 4), torch.tensor([[0, 1, 2, 3]] * 3), [5, 6, 7], 2)
    assert t.indices.dtype == torch.int32 and t.spatial_size == 210 and t.batch_size == 2
    u = t.replace_feature(torch.ones(3, 8))
    assert u.indice_dict is t.indice_dict and u.features.shape[1] == 8 and t.features.shape[1] == 4
    t.features = torch.ones(3, 2)                                            # spconv 1.x style assignment
    assert t.features.shape[1] == 2
    seq = spx.SparseSequential(spx.SubMConv3d(4, 8, 3, indice_key="a"), torch.nn.BatchNorm1d(8), torch.nn.ReLU())
    assert len(seq) == 3 and isinstance(seq[0], spx.conv.SparseConvolution)
    with pytest.raises(NotImplementedError):
        spx.SparseConvolution(2, 4, 4)
    with pytest.raises(NotImplementedError):
        spx.SubMConv3d(4, 4, 3, groups=2)


def test_data_processor_host_steps():
    """DataProcessor (reference data_processor.py:64-226) without a GPU: range mask, box mask, shuffle, grid size."""
    from pcdet_amd.config import AttrDict, cfg_from_yaml_file
    from pcdet_amd.datasets.processor.data_processor import DataProcessor
    from pcdet_amd.utils import box_utils
    cfg = cfg_from_yaml_file(os.path.join(CFG, "kitti_models", "second.yaml"), AttrDict())
    pcr = cfg.DATA_CONFIG.POINT_CLOUD_RANGE
    steps = [AttrDict(NAME="mask_points_and_boxes_outside_range", REMOVE_OUTSIDE_BOXES=True),
             AttrDict(NAME="shuffle_points", SHUFFLE_ENABLED={"train": True, "test": False}),
             AttrDict(NAME="transform_points_to_voxels_placeholder", VOXEL_SIZE=[0.05, 0.05, 0.1])]
    dp = DataProcessor(steps, pcr, training=True, num_point_features=4)
    assert list(dp.grid_size) == [1408, 1600, 40] and dp.voxel_size == [0.05, 0.05, 0.1]
    pts = np.array([[1.0, 0.0, 0.0, 0.5], [70.4, 0.0, 0.0, 0.5], [70.3997, 39.9997, 0.9997, 0.1], [-1.0, 0.0, 0.0, 0.2],
                    [5.0, -40.0, -3.0, 0.3]], np.float32)
    boxes = np.array([[10, 0, -1, 4, 2, 1.5, 0.3, 1], [200, 0, -1, 4, 2, 1.5, 0.0, 1], [70.0, 39.5, 0, 4, 2, 1.5, 0.7, 2]],
                     np.float32)
    np.random.seed(0)
    out = dp.forward({"points": pts.copy(), "gt_boxes": boxes.copy(), "use_lead_xyz": True})
    kept = out["points"]
    assert kept.shape[0] == 3 and set(map(tuple, kept.tolist())) == set(map(tuple, pts[[0, 2, 4]].tolist()))
    assert out["gt_boxes"].shape[0] == 2 and out["gt_boxes"][0, 0] == 10 and out["gt_boxes"][1, 0] == 70.0
    c = box_utils.boxes_to_corners_3d(np.array([[0, 0, 0, 2, 4, 6, np.pi / 2]], np.float32))[0]
    assert np.allclose(c[0], [-2, 1, -3], atol=1e-5) and np.allclose(c[6], [2, -1, 3], atol=1e-5)   # (+x,+y,-z) rotated 90 deg
    test_dp = DataProcessor(steps, pcr, training=False, num_point_features=4)
    out2 = test_dp.forward({"points": pts.copy(), "gt_boxes": boxes.copy(), "use_lead_xyz": True})
    assert np.array_equal(out2["points"], pts[[0, 2, 4]]) and out2["gt_boxes"].shape[0] == 3     # no shuffle, boxes kept


def test_bench_launcher_refuses_what_it_cannot_run():
    """`bench.py --gpus N` must start N ranks itself or fail loudly — never time one rank and call it N (round-1 bug: the
    flag was parsed and ignored).  No GPU here, so both refusals are exercised: fewer GPUs than ranks, and a torchrun
    environment whose WORLD_SIZE disagrees with --gpus.  Neither path touches the GPU."""
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "only" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""


def test_bn_momentum_none_is_not_fused():
    """nn.BatchNorm with momentum=None keeps a cumulative average; the fused kernels take a fixed factor, so such a module
    must stay on the torch path."""
    import torch
    from spx import functional as F_
    bn = torch.nn.BatchNorm1d(16, momentum=None)
    assert not F_.bn_momentum_ok(bn) and F_.bn_momentum_ok(torch.nn.BatchNorm1d(16, momentum=0.01))
    x = torch.randn(8, 16)
    assert not F_.bn_train_fusable(bn, x)
