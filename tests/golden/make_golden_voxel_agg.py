"""Generate tests/golden/ref_voxel_agg.npz from the REFERENCE's pcdet/utils/voxel_aggregation_utils.py (build container
only; same rules as make_golden_ref.py: only data is written, placeholder modules stand in for SharedArray / spconv).

Captured (the fork's voxel-centroid aggregation):
  get_overlapping_voxel_indices (:9-45), get_centroid_per_voxel (:132-161) with and without `num_points_in_voxel`.

Run:  python tests/golden/make_golden_voxel_agg.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_ref import _pkg, _placeholder  # noqa: E402


def main():
    for n in ("SharedArray", "spconv", "spconv.pytorch"):
        _placeholder(n)
    for n in ("pcdet", "pcdet.utils"):
        _pkg(n)
    agg = importlib.import_module("pcdet.utils.voxel_aggregation_utils")
    g = torch.Generator().manual_seed(77)
    out = {}
    pcr = [0.0, -8.0, -3.0, 12.8, 8.0, 1.0]
    vs = [0.05, 0.05, 0.1]
    n = 4000
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([14.0, 18.0, 5.0]) + torch.tensor([-0.5, -9.0, -3.5])
    b = torch.randint(0, 3, (n, 1), generator=g).float()
    feat = torch.rand(n, 2, generator=g)
    points = torch.cat([b, xyz, feat], 1)
    for ds in (1, 4):
        vi = agg.get_overlapping_voxel_indices(points[:, 1:4].clone(), downsample_times=ds, voxel_size=vs,
                                               point_cloud_range=pcr)
        out["ovi_ds%d" % ds] = vi.numpy()
    vi = agg.get_overlapping_voxel_indices(points[:, 1:4].clone(), downsample_times=4, voxel_size=vs, point_cloud_range=pcr)
    vidx = torch.cat((points[:, 0:1].long(), vi), dim=-1)
    ok = (vidx != -1).all(-1)
    vidx_ok, pts_ok = vidx[ok], points[ok]
    cen, cidx, cnt, inv = agg.get_centroid_per_voxel(pts_ok.clone(), vidx_ok.clone())
    out.update(points=points.numpy(), pcr=np.array(pcr, np.float32), vs=np.array(vs, np.float32), cpv_points=pts_ok.numpy(),
               cpv_vidx=vidx_ok.numpy(), cpv_centroids=cen.numpy(), cpv_idx=cidx.numpy(), cpv_count=cnt.numpy(),
               cpv_inverse=inv.numpy())
    # second level: centroids of centroids, weighted by their point counts (what get_centroids_per_voxel_layer intends)
    v2 = cidx.clone()
    v2[:, 1:] = cidx[:, 1:] // 2
    cen2, cidx2, cnt2, inv2 = agg.get_centroid_per_voxel(cen.clone(), v2.clone(), cnt.clone())
    out.update(cpv2_vidx=v2.numpy(), cpv2_centroids=cen2.numpy(), cpv2_idx=cidx2.numpy(), cpv2_count=cnt2.numpy(),
               cpv2_inverse=inv2.numpy())
    np.savez_compressed(os.path.join(HERE, "ref_voxel_agg.npz"), **out)
    print("wrote ref_voxel_agg.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
