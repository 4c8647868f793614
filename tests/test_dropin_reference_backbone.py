"""Boundary B2 drop-in: the REFERENCE's unmodified VoxelBackBone8x / HeightCompression source files, imported with our
`spconv` compat alias (tsm-det-pointcloud-_amd/compat) in place of the spconv wheel, must build the same state_dict and
produce the same outputs as our module.  Needs /root/reference, so it runs in the build container only (skipped on the
GPU box, where the reference does not exist); sparse ops go through the oracle backend here (no GPU)."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pcdet")), reason="reference checkout not present")


def _import_reference_backbone():
    compat = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    import spconv.pytorch  # noqa: F401  (our alias)
    saved = {k: v for k, v in sys.modules.items() if k == "pcdet" or k.startswith("pcdet.")}
    for k in saved:
        del sys.modules[k]
    try:
        sys.modules.setdefault("SharedArray", types.ModuleType("SharedArray"))
        for n in ("pcdet", "pcdet.utils", "pcdet.models", "pcdet.models.backbones_3d", "pcdet.models.backbones_2d",
                  "pcdet.models.backbones_2d.map_to_bev"):
            m = types.ModuleType(n)
            m.__path__ = [os.path.join(REF, *n.split("."))]
            sys.modules[n] = m
        bb = importlib.import_module("pcdet.models.backbones_3d.spconv_backbone")
        hc = importlib.import_module("pcdet.models.backbones_2d.map_to_bev.height_compression")
        return bb, hc
    finally:
        for k in [k for k in sys.modules if k == "pcdet" or k.startswith("pcdet.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_reference_voxelbackbone8x_runs_on_spx():
    from oracle.cpu_backend import use_oracle_backend
    from pcdet_amd.config import AttrDict
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models.backbones_2d.map_to_bev import HeightCompression
    from pcdet_amd.models.backbones_3d import VoxelBackBone8x
    ref_bb, ref_hc = _import_reference_backbone()
    ds = SyntheticDataset(cfg_id=0)
    torch.manual_seed(0)
    ours = VoxelBackBone8x(AttrDict(), 4, ds.grid_size)
    theirs = ref_bb.VoxelBackBone8x(AttrDict(), 4, ds.grid_size)
    assert list(theirs.state_dict().keys()) == list(ours.state_dict().keys())
    assert [tuple(v.shape) for v in theirs.state_dict().values()] == [tuple(v.shape) for v in ours.state_dict().values()]
    theirs.load_state_dict(ours.state_dict())
    ours.eval()
    theirs.eval()
    b = ds.collate_batch([ds[0], ds[1]])
    pts = torch.from_numpy(b["points"])
    with torch.no_grad(), use_oracle_backend():
        from spx import ops
        vox = ops.voxelize(pts, ds.point_cloud_range, ds.voxel_size, 5, 16000, batch_size=2, batch_col=0, xyz_col=1,
                           feat_col=1, want_voxels=False)
        bd = {"voxel_features": vox["mean"], "voxel_coords": vox["coords"].float(), "batch_size": 2}
        out_o = HeightCompression(AttrDict(NUM_BEV_FEATURES=256))(ours(dict(bd)))
        out_t = ref_hc.HeightCompression(AttrDict(NUM_BEV_FEATURES=256))(theirs(dict(bd)))
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        a, c = out_t["multi_scale_3d_features"][k], out_o["multi_scale_3d_features"][k]
        assert torch.equal(a.indices, c.indices) and torch.equal(a.features, c.features)
    assert torch.equal(out_t["spatial_features"], out_o["spatial_features"])
    assert out_t["spatial_features"].shape == (2, 256, 40, 32)
    assert out_t["encoded_spconv_tensor_stride"] == 8


def test_reference_voxelresbackbone8x_runs_on_spx():
    """Row f-3: the reference's residual backbone (SparseBasicBlock, bias=True convs, 128-wide stage) on our operators."""
    from oracle.cpu_backend import use_oracle_backend
    from pcdet_amd.config import AttrDict
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models.backbones_3d import VoxelResBackBone8x
    ref_bb, _ = _import_reference_backbone()
    ds = SyntheticDataset(cfg_id=0)
    torch.manual_seed(1)
    ours = VoxelResBackBone8x(AttrDict(), 4, ds.grid_size)
    theirs = ref_bb.VoxelResBackBone8x(AttrDict(), 4, ds.grid_size)
    assert list(theirs.state_dict().keys()) == list(ours.state_dict().keys())
    theirs.load_state_dict(ours.state_dict())
    ours.eval()
    theirs.eval()
    b = ds.collate_batch([ds[0]])
    with torch.no_grad(), use_oracle_backend():
        from spx import ops
        vox = ops.voxelize(torch.from_numpy(b["points"]), ds.point_cloud_range, ds.voxel_size, 5, 16000, batch_size=1,
                           batch_col=0, xyz_col=1, feat_col=1, want_voxels=False)
        bd = {"voxel_features": vox["mean"], "voxel_coords": vox["coords"].float(), "batch_size": 1}
        o, t = ours(dict(bd)), theirs(dict(bd))
    for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        assert torch.equal(t["multi_scale_3d_features"][k].features, o["multi_scale_3d_features"][k].features)
    assert torch.equal(t["encoded_spconv_tensor"].features, o["encoded_spconv_tensor"].features)
    assert o["encoded_spconv_tensor"].features.shape[1] == 128


def test_reference_unetv2_runs_on_spx():
    """Row f-3: the reference's UNetV2 source (SparseInverseConv3d, 128->64 SubM, k=1 SparseConv3d) on our operators,
    against our own UNetV2: same state_dict, bitwise the same outputs."""
    from oracle.cpu_backend import use_oracle_backend
    from pcdet_amd.config import AttrDict
    from pcdet_amd.datasets import SyntheticDataset
    from pcdet_amd.models.backbones_3d import UNetV2
    _import_reference_backbone()           # primes the alias packages
    compat = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "compat")
    assert compat in sys.path
    saved = {k: v for k, v in sys.modules.items() if k == "pcdet" or k.startswith("pcdet.")}
    for k in saved:
        del sys.modules[k]
    try:
        for n in ("pcdet", "pcdet.utils", "pcdet.models", "pcdet.models.backbones_3d"):
            m = types.ModuleType(n)
            m.__path__ = [os.path.join(REF, *n.split("."))]
            sys.modules[n] = m
        ref_unet = importlib.import_module("pcdet.models.backbones_3d.spconv_unet")
    finally:
        for k in [k for k in sys.modules if k == "pcdet" or k.startswith("pcdet.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    ds = SyntheticDataset(cfg_id=0)
    torch.manual_seed(2)
    kw = dict(voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range)
    ours = UNetV2(AttrDict(), 4, ds.grid_size, **kw)
    theirs = ref_unet.UNetV2(AttrDict(), 4, np.asarray(ds.grid_size), **kw)   # the reference adds [1, 0, 0] to an ndarray
    assert list(theirs.state_dict().keys()) == list(ours.state_dict().keys())
    assert [tuple(v.shape) for v in theirs.state_dict().values()] == [tuple(v.shape) for v in ours.state_dict().values()]
    theirs.load_state_dict(ours.state_dict())
    ours.eval()
    theirs.eval()
    b = ds.collate_batch([ds[0]])
    with torch.no_grad(), use_oracle_backend():
        from spx import ops
        vox = ops.voxelize(torch.from_numpy(b["points"]), ds.point_cloud_range, ds.voxel_size, 5, 16000, batch_size=1,
                           batch_col=0, xyz_col=1, feat_col=1, want_voxels=False)
        bd = {"voxel_features": vox["mean"], "voxel_coords": vox["coords"].float(), "batch_size": 1}
        o, t = ours(dict(bd)), theirs(dict(bd))
    assert o["point_features"].shape == (vox["coords"].shape[0], 16)
    assert torch.equal(t["point_features"], o["point_features"])
    assert torch.equal(t["raw_points_bxyz"], o["raw_points_bxyz"])
    assert torch.equal(t["encoded_spconv_tensor"].features, o["encoded_spconv_tensor"].features)
    assert torch.equal(t["encoded_spconv_tensor"].indices, o["encoded_spconv_tensor"].indices)
