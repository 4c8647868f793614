"""Boundary B1 drop-in: the REFERENCE's own training-loop function — tools/train_utils/train_utils.py:train_one_epoch,
imported unmodified from /root/reference — drives OUR detector through the `pcdet` import alias
(tsm-det-pointcloud-_amd/compat/pcdet): `from pcdet.utils import common_utils, commu_utils` at its top resolves to
pcdet_amd, and the model function is our `pcdet.models.model_fn_decorator()` (reference pcdet/models/__init__.py:40-52:
load_data_to_gpu -> model(batch) -> ModelReturn(loss, tb_dict, disp_dict) -> update_global_step).  Two iterations of
lr_scheduler.step / zero_grad / model_func / backward / clip_grad_norm_ / optimizer.step on the oracle backend (no GPU
here).  Needs /root/reference: build container only."""
import functools
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "tools/train_utils/train_utils.py")),
                                reason="reference checkout not present")


def _alias():
    compat = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    import pcdet
    return pcdet


def test_pcdet_alias_is_the_same_package():
    pcdet = _alias()
    import pcdet_amd
    import pcdet.models.detectors as a
    import pcdet_amd.models.detectors as b
    from pcdet.models import build_network, load_data_to_gpu, model_fn_decorator      # noqa: F401  (tools/train.py:14-17)
    from pcdet.utils import common_utils, commu_utils
    assert pcdet is pcdet_amd and a is b and a.SECONDNet is b.SECONDNet
    assert sys.modules["pcdet.models.detectors"] is sys.modules["pcdet_amd.models.detectors"]
    assert commu_utils.average_reduce_value(3.0) == 3.0 and commu_utils.get_world_size() == 1
    m = common_utils.AverageMeter()
    m.update(2.0), m.update(4.0)
    assert m.avg == 3.0 and m.val == 4.0


def test_reference_train_one_epoch_drives_our_detector(monkeypatch):
    _alias()
    import tqdm
    from oracle.cpu_backend import use_oracle_backend
    import pcdet.models as pm
    from pcdet.config import AttrDict, cfg_from_yaml_file
    from pcdet.datasets import build_dataloader
    from tools.train_utils.optimization import build_optimizer, build_scheduler
    spec = importlib.util.spec_from_file_location("ref_train_utils", os.path.join(REF, "tools/train_utils/train_utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)                      # its `from pcdet.utils import common_utils, commu_utils` = ours

    cfg = cfg_from_yaml_file(os.path.join(ROOT, "tsm-det-pointcloud-_amd/tools/cfgs/kitti_models/second.yaml"), AttrDict())
    cfg.DATA_CONFIG.SYNTHETIC_CFG_ID = 0
    ds, loader, _sampler = build_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, batch_size=2, dist=False, workers=0,
                                            training=True, length=4)
    torch.manual_seed(0)
    model = pm.build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(optimizer, total_iters_each_epoch=2, total_epochs=1, last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    # no GPU in this container: load_data_to_gpu targets the host, sparse ops run on the oracle backend
    monkeypatch.setattr(pm, "load_data_to_gpu", functools.partial(pm.load_data_to_gpu, device=torch.device("cpu")))
    model_func = pm.model_fn_decorator()
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    seen = []

    class _Tb(object):
        def add_scalar(self, key, val, it):
            seen.append((key, float(val), it))

    with use_oracle_backend(), tqdm.trange(0, 1, desc="epochs", disable=True) as tbar:
        it = ref.train_one_epoch(model, optimizer, loader, model_func, lr_scheduler=sched, accumulated_iter=0,
                                 optim_cfg=cfg.OPTIMIZATION, rank=0, tbar=tbar, total_it_each_epoch=len(loader),
                                 dataloader_iter=iter(loader), tb_log=_Tb())
    assert it == 2 and int(model.global_step) == 2            # update_global_step ran inside model_func
    losses = [v for k, v, _ in seen if k == "train/loss"]
    assert len(losses) == 2 and all(np.isfinite(losses))
    assert {"train/rpn_loss_cls", "train/rpn_loss_loc", "train/rpn_loss_dir", "meta_data/learning_rate"} <= {k for k, _v, _i in seen}
    moved = [k for k, v in model.named_parameters() if not torch.equal(v, before[k])]
    assert any(k.startswith("backbone_3d.conv_input.0") for k in moved) and any(k.startswith("dense_head") for k in moved)
    assert model.training
