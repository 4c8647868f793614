"""build_network / load_data_to_gpu / model_fn_decorator (reference pcdet/models/__init__.py:16-52)."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict, device=None):
    """ndarray -> float tensor on the GPU (voxel_coords stay float here and are cast with .int() in the backbone,
    exactly like the reference); tensors already on the device are left alone."""
    device = torch.device('cuda') if device is None else device
    for key, val in batch_dict.items():
        if isinstance(val, np.ndarray):
            if key in ('frame_id', 'metadata', 'calib', 'image_shape'):
                continue
            batch_dict[key] = torch.from_numpy(val).float().to(device, non_blocking=True)
        elif isinstance(val, torch.Tensor) and val.device != device:
            batch_dict[key] = val.to(device, non_blocking=True)
    return batch_dict


def model_fn_decorator():
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss'].mean()
        if hasattr(model, 'update_global_step'):
            model.update_global_step()
        else:
            model.module.update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
