"""Dev tool (GPU box): the rule-table builds of one batch in a loop, for `rocprofv3 --kernel-trace --stats` (per-kernel
averages of the index kernels).  python tools/rulebook_prof.py [--cfg 2] [--iters 30]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from pcdet_amd.datasets import synthetic  # noqa: E402
from spx import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", type=int, default=2)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
spec = synthetic.CONFIGS[a.cfg]
geom, batch = spec["geom"], spec["batch"]
pts = torch.from_numpy(synthetic.make_batch(a.cfg, batch)["points"]).to(dev)
mv = spec.get("max_voxels", geom["max_voxels"]["train"])
gs = synthetic.grid_size_of(geom)
chain = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
for it in range(a.iters):
    vox = ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, mv, batch_size=batch, batch_col=0, xyz_col=1,
                       feat_col=1, want_voxels=False)
    idx, shape = vox["coords"], [int(gs[2]) + 1, int(gs[1]), int(gs[0])]
    ops.subm_rulebook(idx, batch, shape, (3, 3, 3))
    for j, (k, s, p) in enumerate(chain):
        rb = ops.conv_rulebook(idx, batch, shape, k, s, p, subm_ksize=(3, 3, 3) if j < 3 else None)
        idx, shape = rb.out_indices, rb.out_shape
torch.cuda.synchronize()
print("done", a.cfg, a.iters)
