"""Anchor grid (semantics of reference pcdet/models/dense_heads/target_assigner/anchor_generator.py:4-60).
Built on the CPU (no hard-coded .cuda()); the head registers the result as buffers so .to(device) moves it."""
import torch


class AnchorGenerator(object):
    def __init__(self, anchor_range, anchor_generator_config):
        self.anchor_generator_cfg = anchor_generator_config
        self.anchor_range = [float(v) for v in anchor_range]
        self.anchor_sizes = [c['anchor_sizes'] for c in anchor_generator_config]
        self.anchor_rotations = [c['anchor_rotations'] for c in anchor_generator_config]
        self.anchor_heights = [c['anchor_bottom_heights'] for c in anchor_generator_config]
        self.align_center = [c.get('align_center', False) for c in anchor_generator_config]
        assert len(self.anchor_sizes) == len(self.anchor_rotations) == len(self.anchor_heights)
        self.num_of_anchor_sets = len(self.anchor_sizes)

    def generate_anchors(self, grid_sizes):
        """grid_sizes: per class (nx, ny).  Returns ([1(z), ny, nx, n_size, n_rot, 7] per class, anchors/location)."""
        assert len(grid_sizes) == self.num_of_anchor_sets
        r = self.anchor_range
        all_anchors, per_location = [], []
        for grid_size, sizes, rots, heights, centre in zip(grid_sizes, self.anchor_sizes, self.anchor_rotations,
                                                           self.anchor_heights, self.align_center):
            per_location.append(len(rots) * len(sizes) * len(heights))
            nx, ny = int(grid_size[0]), int(grid_size[1])
            if centre:
                sx, sy = (r[3] - r[0]) / nx, (r[4] - r[1]) / ny
                ox, oy = sx / 2, sy / 2
            else:
                sx, sy = (r[3] - r[0]) / (nx - 1), (r[4] - r[1]) / (ny - 1)
                ox, oy = 0, 0
            xs = torch.arange(r[0] + ox, r[3] + 1e-5, step=sx, dtype=torch.float32)
            ys = torch.arange(r[1] + oy, r[4] + 1e-5, step=sy, dtype=torch.float32)
            zs = torch.tensor(heights, dtype=torch.float32)
            sizes_t = torch.tensor(sizes, dtype=torch.float32)     # [n_size, 3]
            rots_t = torch.tensor(rots, dtype=torch.float32)       # [n_rot]
            gx, gy, gz = torch.meshgrid(xs, ys, zs, indexing='ij')  # [x, y, z]
            n_size, n_rot = sizes_t.shape[0], rots_t.shape[0]
            shape = (*gx.shape, n_size, n_rot)
            a = torch.empty(*shape, 7, dtype=torch.float32)
            a[..., 0] = gx[..., None, None]
            a[..., 1] = gy[..., None, None]
            a[..., 2] = gz[..., None, None]
            a[..., 3:6] = sizes_t.view(1, 1, 1, n_size, 1, 3)
            a[..., 6] = rots_t.view(1, 1, 1, 1, n_rot)
            a = a.permute(2, 1, 0, 3, 4, 5).contiguous()           # [z, y, x, size, rot, 7]
            a[..., 2] += a[..., 5] / 2                             # bottom height -> box centre
            all_anchors.append(a)
        return all_anchors, per_location
