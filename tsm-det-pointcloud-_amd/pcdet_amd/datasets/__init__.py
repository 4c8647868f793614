"""Dataset objects.  Only the synthetic generator is in scope (real loaders are out of scope, real data absent).
`SyntheticDataset` exposes the attributes Detector3DTemplate.build_networks reads
(reference pcdet/models/detectors/detector3d_template.py:39-47): class_names, point_feature_encoder.num_point_features,
grid_size, point_cloud_range, voxel_size, depth_downsample_factor."""
from types import SimpleNamespace

import numpy as np
import torch
from torch.utils.data import Dataset

from . import synthetic


class SyntheticDataset(Dataset):
    def __init__(self, dataset_cfg=None, class_names=None, training=True, cfg_id=None, length=64, **kwargs):
        self.dataset_cfg = dataset_cfg
        self.training = training
        self.cfg_id = int(cfg_id if cfg_id is not None else dataset_cfg.get('SYNTHETIC_CFG_ID', 2))
        spec = synthetic.CONFIGS[self.cfg_id]
        geom = spec['geom']
        self.class_names = list(class_names) if class_names is not None else list(geom['class_names'])
        self.point_cloud_range = np.array(geom['point_cloud_range'], dtype=np.float32)
        self.voxel_size = list(geom['voxel_size'])
        self.grid_size = synthetic.grid_size_of(geom)          # (gx, gy, gz), data_processor.py:129-130
        self.point_feature_encoder = SimpleNamespace(num_point_features=geom['num_point_features'])
        self.depth_downsample_factor = None
        self.max_points_per_voxel = geom['max_points_per_voxel']
        self.max_voxels = dict(geom['max_voxels'])
        if 'max_voxels' in spec:
            self.max_voxels = {'train': spec['max_voxels'], 'test': spec['max_voxels']}
        self.length = length

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        f = synthetic.make_frame(self.cfg_id, index)
        return {'points': f['points'], 'gt_boxes': f['gt_boxes'], 'frame_id': index}

    @staticmethod
    def collate_batch(batch_list, _unused=False):
        """points -> [N, 1+C] with the frame index in column 0; gt_boxes padded to [B, Mmax, 8]
        (reference pcdet/datasets/dataset.py:161-229)."""
        pts = [np.pad(s['points'], ((0, 0), (1, 0)), mode='constant', constant_values=i)
               for i, s in enumerate(batch_list)]
        mmax = max(s['gt_boxes'].shape[0] for s in batch_list)
        gt = np.zeros((len(batch_list), mmax, batch_list[0]['gt_boxes'].shape[-1]), np.float32)
        for i, s in enumerate(batch_list):
            gt[i, :s['gt_boxes'].shape[0]] = s['gt_boxes']
        return {'points': np.concatenate(pts, 0), 'gt_boxes': gt, 'batch_size': len(batch_list),
                'frame_id': np.array([s['frame_id'] for s in batch_list])}


__all__ = {'SyntheticDataset': SyntheticDataset}


def build_dataloader(dataset_cfg, class_names, batch_size, dist=False, workers=0, training=True, total_epochs=0,
                     length=64, **kwargs):
    """(dataset, dataloader, sampler) as the reference's build_dataloader returns (pcdet/datasets/__init__.py:47-76).
    Frames are sharded across ranks with DistributedSampler (rank::world_size in eval)."""
    dataset = __all__[dataset_cfg.DATASET](dataset_cfg=dataset_cfg, class_names=class_names, training=training,
                                           length=length)
    sampler = None
    if dist:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=training)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, pin_memory=True, num_workers=workers,
                                         shuffle=(sampler is None) and training, collate_fn=dataset.collate_batch,
                                         drop_last=False, sampler=sampler, timeout=0)
    return dataset, loader, sampler
