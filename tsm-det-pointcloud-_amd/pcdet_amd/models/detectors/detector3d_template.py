"""Detector3DTemplate — module assembly, checkpoint IO and post-processing with the reference's API
(pcdet/models/detectors/detector3d_template.py:17-205 build_*, :207-349 post_processing, :544-625 checkpoints).

Differences forced by the fork's drift, all backwards compatible:
  * build_backbone_2d keeps `num_bev_features` from the module's own attribute (falls back to the fork's
    num_voxel_neck_features) and never overwrites num_point_features with None;
  * POST_PROCESSING.SCORE_THRESH may be a scalar (upstream) or a per-class list (fork's multi_thresh).
Only the module slots of the SECOND path are populated; the other slots of `module_topology` stay None.
"""
import os

import torch
import torch.nn as nn

from ...utils.spconv_utils import find_all_spconv_keys
from .. import backbones_2d, backbones_3d, dense_heads
from ..backbones_2d import map_to_bev
from ..backbones_3d import vfe
from ..model_utils import model_nms_utils


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['vfe', 'backbone_3d', 'map_to_bev_module', 'pfe', 'backbone_2d', 'neck', 'dense_head',
                                'point_head', 'roi_head']

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {
            'module_list': [],
            'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
            'num_point_features': self.dataset.point_feature_encoder.num_point_features,
            'grid_size': self.dataset.grid_size,
            'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size,
            'depth_downsample_factor': getattr(self.dataset, 'depth_downsample_factor', None),
        }
        for module_name in self.module_topology:
            module, info = getattr(self, 'build_%s' % module_name)(model_info_dict=info)
            self.add_module(module_name, module)
        return info['module_list']

    def build_vfe(self, model_info_dict):
        if self.model_cfg.get('VFE', None) is None:
            return None, model_info_dict
        m = vfe.__all__[self.model_cfg.VFE.NAME](
            model_cfg=self.model_cfg.VFE, num_point_features=model_info_dict['num_rawpoint_features'],
            point_cloud_range=model_info_dict['point_cloud_range'], voxel_size=model_info_dict['voxel_size'],
            grid_size=model_info_dict['grid_size'], depth_downsample_factor=model_info_dict['depth_downsample_factor'])
        model_info_dict['num_point_features'] = m.get_output_feature_dim()
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_backbone_3d(self, model_info_dict):
        if self.model_cfg.get('BACKBONE_3D', None) is None:
            return None, model_info_dict
        m = backbones_3d.__all__[self.model_cfg.BACKBONE_3D.NAME](
            model_cfg=self.model_cfg.BACKBONE_3D, input_channels=model_info_dict['num_point_features'],
            grid_size=model_info_dict['grid_size'], voxel_size=model_info_dict['voxel_size'],
            point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_point_features'] = m.num_point_features
        model_info_dict['backbone_channels'] = getattr(m, 'backbone_channels', None)
        model_info_dict['num_bev_features'] = getattr(m, 'num_bev_features', None)
        return m, model_info_dict

    def build_map_to_bev_module(self, model_info_dict):
        if self.model_cfg.get('MAP_TO_BEV', None) is None:
            return None, model_info_dict
        m = map_to_bev.__all__[self.model_cfg.MAP_TO_BEV.NAME](model_cfg=self.model_cfg.MAP_TO_BEV,
                                                               grid_size=model_info_dict['grid_size'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        if self.model_cfg.get('BACKBONE_2D', None) is None:
            return None, model_info_dict
        m = backbones_2d.__all__[self.model_cfg.BACKBONE_2D.NAME](
            model_cfg=self.model_cfg.BACKBONE_2D, input_channels=model_info_dict['num_bev_features'],
            voxel_size=model_info_dict['voxel_size'], point_cloud_range=model_info_dict['point_cloud_range'],
            backbone_channels=model_info_dict.get('backbone_channels', None))
        model_info_dict['module_list'].append(m)
        nbev = getattr(m, 'num_bev_features', None)
        model_info_dict['num_bev_features'] = nbev if nbev is not None else m.num_voxel_neck_features
        if getattr(m, 'num_point_features', None) is not None:
            model_info_dict['num_point_features'] = m.num_point_features
        return m, model_info_dict

    def build_dense_head(self, model_info_dict):
        if self.model_cfg.get('DENSE_HEAD', None) is None:
            return None, model_info_dict
        m = dense_heads.__all__[self.model_cfg.DENSE_HEAD.NAME](
            model_cfg=self.model_cfg.DENSE_HEAD, input_channels=model_info_dict['num_bev_features'],
            num_class=self.num_class if not self.model_cfg.DENSE_HEAD.CLASS_AGNOSTIC else 1,
            class_names=self.class_names, grid_size=model_info_dict['grid_size'],
            point_cloud_range=model_info_dict['point_cloud_range'],
            predict_boxes_when_training=self.model_cfg.get('ROI_HEAD', False),
            voxel_size=model_info_dict.get('voxel_size', False))
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def _absent(self, key, model_info_dict):
        if self.model_cfg.get(key, None) is not None:
            raise NotImplementedError('%s modules are outside the SECOND hot path (SURVEY.md §2)' % key)
        return None, model_info_dict

    def build_pfe(self, model_info_dict):
        return self._absent('PFE', model_info_dict)

    def build_neck(self, model_info_dict):
        return self._absent('NECK', model_info_dict)

    def build_point_head(self, model_info_dict):
        return self._absent('POINT_HEAD', model_info_dict)

    def build_roi_head(self, model_info_dict):
        return self._absent('ROI_HEAD', model_info_dict)

    def forward(self, **kwargs):
        raise NotImplementedError

    # ------------------------------------------------------------------ post-processing (reference :207-349)
    def post_processing(self, batch_dict):
        cfg = self.model_cfg.POST_PROCESSING
        batch_size = batch_dict['batch_size']
        recall_dict, pred_dicts = {}, []
        for index in range(batch_size):
            box_preds = batch_dict['batch_box_preds'][index]
            cls_preds = batch_dict['batch_cls_preds'][index]
            src_cls_preds = cls_preds
            assert cls_preds.shape[1] in [1, self.num_class]
            if not batch_dict['cls_preds_normalized']:
                cls_preds = torch.sigmoid(cls_preds)
            if cfg.NMS_CONFIG.MULTI_CLASSES_NMS:
                raise NotImplementedError('MULTI_CLASSES_NMS is not used by the SECOND configuration')
            cls_preds, label_preds = torch.max(cls_preds, dim=-1)
            label_preds = label_preds + 1
            thresh = cfg.SCORE_THRESH
            if isinstance(thresh, (list, tuple)):   # fork: per-class thresholds + a final cross-class NMS
                selected, selected_scores = model_nms_utils.multi_thresh(
                    box_scores=cls_preds, box_labels=label_preds, box_preds=box_preds, nms_config=cfg.NMS_CONFIG,
                    score_thresh=thresh)
            else:                                   # upstream: one class-agnostic NMS
                selected, selected_scores = model_nms_utils.class_agnostic_nms(
                    box_scores=cls_preds, box_preds=box_preds, nms_config=cfg.NMS_CONFIG, score_thresh=thresh)
            if cfg.get('OUTPUT_RAW_SCORE', False):
                selected_scores = torch.max(src_cls_preds, dim=-1)[0][selected]
            pred_dicts.append({'pred_boxes': box_preds[selected], 'pred_scores': selected_scores,
                               'pred_labels': label_preds[selected]})
        return pred_dicts, recall_dict

    # ------------------------------------------------------------------ checkpoints (reference :544-625)
    def _load_state_dict(self, model_state_disk, *, strict=True):
        state_dict = self.state_dict()
        spconv_keys = find_all_spconv_keys(self)
        update = {}
        for key, val in model_state_disk.items():
            if key in spconv_keys and key in state_dict and state_dict[key].shape != val.shape:
                # spconv 1.x stored (k1,k2,k3,Cin,Cout); ours (= spconv 2.x implicit-gemm) is (Cout,k1,k2,k3,Cin)
                native = val.transpose(-1, -2)
                if native.shape == state_dict[key].shape:
                    val = native.contiguous()
                else:
                    assert val.dim() == 5, 'currently only spconv 3D is supported'
                    implicit = val.permute(4, 0, 1, 2, 3)
                    if implicit.shape == state_dict[key].shape:
                        val = implicit.contiguous()
            if key in state_dict and state_dict[key].shape == val.shape:
                update[key] = val
        if strict:
            self.load_state_dict(update)
        else:
            state_dict.update(update)
            self.load_state_dict(state_dict)
        return state_dict, update

    def load_params_from_file(self, filename, logger, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None, weights_only=True)
        state_dict, update = self._load_state_dict(checkpoint['model_state'], strict=False)
        for key in state_dict:
            if key not in update:
                logger.info('Not updated weight %s: %s' % (key, str(state_dict[key].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(update), len(state_dict)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None, weights_only=True)
        epoch, it = checkpoint.get('epoch', -1), checkpoint.get('it', 0.0)
        self._load_state_dict(checkpoint['model_state'], strict=True)
        if optimizer is not None and checkpoint.get('optimizer_state', None) is not None:
            optimizer.load_state_dict(checkpoint['optimizer_state'])
        return it, epoch
