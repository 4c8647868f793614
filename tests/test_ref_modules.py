"""Our pcdet-API modules against vectors captured from the REFERENCE's own pure-torch modules
(tests/golden/ref_modules.npz, made by tests/golden/make_golden_ref.py in the build container).
CPU-only (the dense tail is torch code on both sides); tolerances are fp32 round-off."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "ref_modules.npz"))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_residual_coder(G):
    from pcdet_amd.utils.box_coder_utils import ResidualCoder
    c = ResidualCoder()
    enc = c.encode_torch(_t(G["coder_boxes"]), _t(G["coder_anchors"]))
    assert np.abs(enc.numpy() - G["coder_enc"]).max() < 1e-6
    dec = c.decode_torch(_t(G["coder_codes"]), _t(G["coder_anchors"])[None].repeat(2, 1, 1))
    assert np.abs(dec.numpy() - G["coder_dec"]).max() < 1e-5
    rt = c.decode_torch(enc, _t(G["coder_anchors"]))
    assert np.abs(rt.numpy() - G["coder_boxes"]).max() < 1e-4


def test_limit_period_and_bev_iou(G):
    from pcdet_amd.utils import box_utils, common_utils
    a = _t(G["lp_in"])
    assert np.array_equal(common_utils.limit_period(a, 0.5, np.pi).numpy(), G["lp_out_a"])
    assert np.array_equal(common_utils.limit_period(a - 0.78539, 0.0, np.pi).numpy(), G["lp_out_b"])
    iou = box_utils.boxes3d_nearest_bev_iou(_t(G["coder_anchors"])[:50], _t(G["coder_boxes"])[:40])
    assert np.abs(iou.numpy() - G["iou_ab"]).max() < 1e-6


def _head(G):
    from pcdet_amd.config import AttrDict, cfg_from_yaml_file
    from pcdet_amd.models.dense_heads import AnchorHeadSingle
    root = os.path.dirname(HERE)
    cfg = cfg_from_yaml_file(os.path.join(root, "tsm-det-pointcloud-_amd/tools/cfgs/kitti_models/second.yaml"), AttrDict())
    head = AnchorHeadSingle(model_cfg=cfg.MODEL.DENSE_HEAD, input_channels=16, num_class=3,
                            class_names=["Car", "Pedestrian", "Cyclist"], grid_size=G["head_grid_size"],
                            point_cloud_range=G["head_pc_range"], predict_boxes_when_training=True)
    sd = {k[5:]: _t(G[k]) for k in G.files if k.startswith("head_conv_")}
    head.load_state_dict(sd, strict=True)   # anchors are non-persistent buffers: exactly the reference's key set
    return head


def test_anchor_generator(G):
    head = _head(G)
    for i, a in enumerate(head.anchors):
        assert a.shape == G["anchors_%d" % i].shape
        assert np.abs(a.numpy() - G["anchors_%d" % i]).max() < 1e-5


def test_anchor_head_forward_targets_losses(G):
    head = _head(G)
    head.train()
    dd = head({"encoded_bev_features": [_t(G["head_feat"])], "gt_boxes": _t(G["head_gt"]), "batch_size": 2})
    fr = head.forward_ret_dict
    for k in ("cls_preds", "box_preds", "dir_cls_preds"):
        assert np.abs(fr[k].detach().numpy() - G["head_" + k]).max() < 1e-5, k
    assert np.array_equal(fr["box_cls_labels"].numpy(), G["head_box_cls_labels"])       # integer labels: exact
    assert np.array_equal(fr["reg_weights"].numpy(), G["head_reg_weights"])
    assert np.abs(fr["box_reg_targets"].numpy() - G["head_box_reg_targets"]).max() < 1e-5
    assert np.abs(dd["batch_cls_preds"].detach().numpy() - G["head_batch_cls_preds"]).max() < 1e-5
    assert np.abs(dd["batch_box_preds"].detach().numpy() - G["head_batch_box_preds"]).max() < 1e-4
    loss, tb = head.get_loss()
    parts = np.array([float(tb["rpn_loss_cls"]), float(tb["rpn_loss_loc"]), float(tb["rpn_loss_dir"])])
    assert np.abs(parts - G["head_loss_parts"]).max() < 1e-5
    assert abs(float(loss) - float(G["head_loss"])) < 1e-5
    # also through the upstream key
    dd2 = head({"spatial_features_2d": _t(G["head_feat"]), "gt_boxes": _t(G["head_gt"]), "batch_size": 2})
    assert torch.equal(dd2["batch_box_preds"], dd["batch_box_preds"])


def test_target_assigner_all_padding(G):
    head = _head(G)
    t = head.assign_targets(torch.zeros(1, 3, 8))
    assert np.array_equal(t["box_cls_labels"].numpy(), G["head_empty_labels"])


def test_base_bev_backbone(G):
    from pcdet_amd.config import AttrDict
    from pcdet_amd.models.backbones_2d import BaseBEVBackbone
    cfg = AttrDict(LAYER_NUMS=[1, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16], UPSAMPLE_STRIDES=[1, 2],
                   NUM_UPSAMPLE_FILTERS=[8, 8])
    # extra kwargs = what the fork's Detector3DTemplate.build_backbone_2d passes (detector3d_template.py:107-113)
    m = BaseBEVBackbone(cfg, 6, voxel_size=[0.05, 0.05, 0.1], point_cloud_range=[0] * 6, backbone_channels={})
    m.load_state_dict({k[7:]: _t(G[k]) for k in G.files if k.startswith("bev_sd_")}, strict=True)
    m.eval()
    with torch.no_grad():
        d = m({"spatial_features": _t(G["bev_in"])})
    assert np.abs(d["spatial_features_2d"].numpy() - G["bev_out"]).max() < 1e-5
    assert d["encoded_bev_features"][0] is d["spatial_features_2d"]
    assert m.num_bev_features == m.num_voxel_neck_features == 16


def test_mean_vfe_oracle_matches_reference(G, orc):
    """Pins the ORACLE's MeanVFE to the reference module's output (the HIP kernel is checked against the oracle)."""
    out = orc.mean_vfe(G["vfe_voxels"], G["vfe_num"].astype(np.int32))
    assert np.abs(out - G["vfe_out"]).max() < 1e-6


def test_state_dict_keys_match_reference_layout():
    """Key names / shapes of the sparse backbone so existing checkpoints load."""
    from pcdet_amd.config import AttrDict
    from pcdet_amd.models.backbones_3d import VoxelBackBone8x
    m = VoxelBackBone8x(AttrDict(), 4, np.array([1408, 1600, 40]))
    sd = m.state_dict()
    assert m.sparse_shape == [41, 1600, 1408]
    assert tuple(sd["conv_input.0.weight"].shape) == (16, 3, 3, 3, 4)
    assert tuple(sd["conv_out.0.weight"].shape) == (128, 3, 1, 1, 64)
    for k in ("conv1.0.0.weight", "conv2.0.0.weight", "conv2.2.1.running_var", "conv4.1.1.num_batches_tracked",
              "conv_out.1.bias"):
        assert k in sd
    n_conv = sum(v.numel() for k, v in sd.items() if k.endswith(".0.weight"))
    assert n_conv == 710592
    from pcdet_amd.utils.spconv_utils import find_all_spconv_keys
    assert len(find_all_spconv_keys(m)) == 12
