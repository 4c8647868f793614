"""hipGraph-captured, host-sync-free inference forward (MI355X-first execution model).

The reference's forward (pcdet/models/detectors/second_net.py:9-22) launches ~150 kernels per batch and — through spconv
— synchronises with the host for every data-dependent size (number of voxels, active outputs of each strided conv).
Here every libspx kernel can read its live row count from device memory (include/spx.h, the `d_n*` arguments), so the
whole path  points -> voxelise -> 12 sparse convs (+ fused BN/ReLU) -> densify -> BEV backbone -> anchor head decode
runs at static CAPACITY shapes without a single host sync and is captured once into a hipGraph
(torch.cuda.CUDAGraph); each call is a copy into the static point buffer plus one graph replay.

Capacities: voxels = min(max_points, batch * MAX_NUMBER_OF_VOXELS['test']); the output rows of each strided conv default
to `level_factors` x the voxel capacity (LiDAR scenes shrink after the first stride-2 stage; the no-overflow bound is 8x
per stage).  Rows beyond a capacity are dropped and reported by `overflowed()` (reads the device counters: one sync, call
it when convenient).
"""
import os

import torch

from spx.functional import refresh_folded_bn, refresh_graph_constants

DEFAULT_LEVEL_FACTORS = {'spconv2': 2.0, 'spconv3': 2.0, 'spconv4': 1.0, 'spconv_down2': 1.0}


def static_caps_for(model, batch_size, n_points, training=None, level_factors=None):
    """Row capacities for the host-sync-free (static-capacity) execution of `model` on batches of at most `n_points`
    points: put the returned dict in batch_dict['static_caps'] (training or eval).  The voxel capacity is exact — the
    reference's hard cap min(points, batch * MAX_NUMBER_OF_VOXELS[mode]); the output rows of each strided conv get
    `level_factors` x that (the no-overflow bound is 8x per stage and would size every later kernel for rows that never
    exist).  An overflow never passes silently: the rule-table kernels raise SPX_ERR_CAPACITY in the device status word
    (spx.ops.check_status) and the live counts stay readable in the tensors' n_valid."""
    training = model.training if training is None else training
    vox_cap = min(int(n_points), int(batch_size) * int(model.vfe.max_voxels['train' if training else 'test']))
    f = dict(DEFAULT_LEVEL_FACTORS)
    f.update(level_factors or {})
    caps = {k: max(1, int(v * vox_cap)) for k, v in f.items()}
    caps['voxels'] = vox_cap
    return caps


class GraphedDetector(object):
    def __init__(self, model, batch_size, max_points, level_factors=None, warmup=2, n_modules=None):
        """n_modules: run only the first n modules of model.module_list (3 = voxelise + VFE, 3-D backbone, BEV collapse)."""
        self.model = model.eval()
        self.modules = list(model.module_list) if n_modules is None else list(model.module_list)[:int(n_modules)]
        self.batch_size = int(batch_size)
        vfe = model.vfe
        self.device = next(model.parameters()).device
        assert self.device.type == 'cuda'
        c = 1 + vfe.num_point_features
        self.max_points = int(max_points)
        self.points = torch.empty((self.max_points, c), dtype=torch.float32, device=self.device)
        self._pad_row = torch.zeros((c,), dtype=torch.float32, device=self.device)
        self._pad_row[0] = self.batch_size - 1                       # keeps frame ids ascending
        self._pad_row[1] = float(vfe.point_cloud_range[0]) - 1.0e4   # far outside the range: dropped by the voxeliser
        self.points[:] = self._pad_row
        vox_cap = min(self.max_points, self.batch_size * int(vfe.max_voxels['test']))
        f = dict(DEFAULT_LEVEL_FACTORS)
        f.update(level_factors or {})
        self.static_caps = {k: max(1, int(v * vox_cap)) for k, v in f.items()}
        self.vox_cap = vox_cap
        self.out = None
        # eager warm-up on a side stream (allocator pools, workspaces, MIOpen kernel selection), then capture
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        dot = os.environ.get("SPX_GRAPH_DOT")           # dev knob: write the captured graph (nodes + dependencies) as DOT
        if dot:
            self.graph.enable_debug_mode()
        with torch.cuda.graph(self.graph):
            self.out = self._forward()
        if dot:
            self.graph.debug_dump(dot)

    def _forward(self):
        bd = {'points': self.points, 'batch_size': self.batch_size, 'static_caps': self.static_caps}
        with torch.no_grad():
            for m in self.modules:
                bd = m(bd)
        ms = bd['multi_scale_3d_features']
        out = {k: bd[k] for k in ('batch_cls_preds', 'batch_box_preds', 'cls_preds_normalized') if k in bd}
        out.update({'batch_size': self.batch_size,
                'spatial_features': bd['spatial_features'],
                'counts': {'voxels': bd['voxel_num_valid'], 'spconv2': ms['x_conv2'].n_valid,
                           'spconv3': ms['x_conv3'].n_valid, 'spconv4': ms['x_conv4'].n_valid,
                           'spconv_down2': bd['encoded_spconv_tensor'].n_valid}})
        return out

    def __call__(self, points):
        """points [N, 1+C] (frame index in column 0, frames contiguous and ascending), N <= max_points, on the GPU.
        Returns the static output dict (overwritten by the next call)."""
        n = points.shape[0]
        if n > self.max_points:
            raise ValueError("%d points > capacity %d" % (n, self.max_points))
        self.points[:n].copy_(points, non_blocking=True)
        if n < self.max_points:
            self.points[n:] = self._pad_row
        refresh_folded_bn(self.model)        # BatchNorm folds the graph captured: rebuilt in place if a parameter was written
        refresh_graph_constants()            # likewise the packed weights / Winograd images it reads
        self.graph.replay()
        return self.out

    def overflowed(self):
        """{level: (live count, capacity)} for every level whose capacity was exceeded in the LAST call (one host sync)."""
        caps = dict(self.static_caps, voxels=self.vox_cap)
        bad = {}
        for k, t in self.out['counts'].items():
            v = int(t.item())
            if v > caps[k]:
                bad[k] = (v, caps[k])
        return bad

    def post_process(self):
        """score threshold + NMS on the last outputs (data dependent: outside the graph)."""
        return self.model.post_processing(dict(self.out))


class GraphedTrainStep(object):
    """Forward + backward of one training step as ONE hipGraph replay (single process; data-parallel ranks stay eager: DDP's
    bucket hooks and the RCCL all-reduce are not captured here).

    The training step issues ~750 kernels; at static row capacities none of them needs the host, so the whole
    forward + backward — voxeliser, index stream, sparse and dense convolutions, losses, every gradient — is captured
    once and replayed per step.  The host then issues a copy of the batch into the static buffers, one replay, gradient
    clipping and the (fused) optimizer step: the reference's loop (tools/train_utils/train_utils.py:44-56: zero_grad,
    model_func, backward, clip_grad_norm_, optimizer.step) with its first three calls folded into the replay.  Parameter
    gradients live in the graph's memory pool and are rewritten by every replay, so there is no zero_grad between steps.

    points [N, 1+C] (frame index in column 0, frames ascending), gt_boxes [B, M, 8]; N <= max_points, M <= max_gt."""

    def __init__(self, model, batch_size, max_points, max_gt, level_factors=None, warmup=3, example=None):
        assert model.training, "GraphedTrainStep captures the training forward"
        self.model = model
        self.batch_size = int(batch_size)
        self.device = next(model.parameters()).device
        vfe = model.vfe
        c = 1 + vfe.num_point_features
        self.max_points, self.max_gt = int(max_points), int(max_gt)
        self.points = torch.empty((self.max_points, c), dtype=torch.float32, device=self.device)
        self._pad_row = torch.zeros((c,), dtype=torch.float32, device=self.device)
        self._pad_row[0] = self.batch_size - 1
        self._pad_row[1] = float(vfe.point_cloud_range[0]) - 1.0e4          # outside the range: dropped by the voxeliser
        self.points[:] = self._pad_row
        self.gt_boxes = torch.zeros((self.batch_size, self.max_gt, 8), dtype=torch.float32, device=self.device)
        self.static_caps = static_caps_for(model, self.batch_size, self.max_points, training=True,
                                           level_factors=level_factors)
        if example is not None:
            self._load(example[0], example[1])
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):                      # allocator pools, workspaces, MIOpen kernel selection
                model.zero_grad(set_to_none=True)
                self._forward_backward()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        model.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._forward_backward()

    def _forward_backward(self):
        bd = {'points': self.points, 'gt_boxes': self.gt_boxes, 'batch_size': self.batch_size,
              'static_caps': self.static_caps}
        ret, tb_dict, _disp = self.model(bd)
        loss = ret['loss'].mean()
        loss.backward()
        return {'loss': loss.detach(), 'tb_dict': tb_dict}

    def _load(self, points, gt_boxes):
        n, m = points.shape[0], gt_boxes.shape[1]
        if n > self.max_points or m > self.max_gt or gt_boxes.shape[0] != self.batch_size:
            raise ValueError("batch (%d points, %d boxes) exceeds the captured capacities (%d, %d)"
                             % (n, m, self.max_points, self.max_gt))
        self.points[:n].copy_(points, non_blocking=True)
        if n < self.max_points:
            self.points[n:] = self._pad_row
        self.gt_boxes.zero_()
        self.gt_boxes[:, :m].copy_(gt_boxes, non_blocking=True)

    def __call__(self, points, gt_boxes):
        """Copies the batch in and replays forward + backward; parameter .grad tensors hold the new gradients afterwards.
        Returns the static output dict {'loss', 'tb_dict'} (overwritten by the next call)."""
        self._load(points, gt_boxes)
        self.graph.replay()
        return self.out
