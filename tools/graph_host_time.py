"""Dev tool (GPU box): host time of one replay of the inference graph against its GPU time (profiles/r03_conv_experiments.md)."""
import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch
import bench
from pcdet_amd.utils.miopen_db import use_tuned_db
use_tuned_db()
dev=torch.device("cuda:0")
cfg, ds, model, opt, sched = bench.build(2, dev, "f32")
model.eval()
batches = bench.make_batches(ds, 2, 4, 0, dev)
from pcdet_amd.models.inference import GraphedDetector
pts=batches[0]["points"]
for pipelined in (False,):        # (the per-stage capture measured in round 3 was removed again)
    r = GraphedDetector(model, 4, int(pts.shape[0]*1.05)+64)
    for _ in range(5): r(pts)
    torch.cuda.synchronize()
    hs=[]; ts=[]
    for _ in range(20):
        t0=time.perf_counter(); r(pts); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        hs.append((t1-t0)*1e3); ts.append((t2-t0)*1e3)
    hs.sort(); ts.sort()
    print("pipelined=%s: host time of a replay call %.3f ms (median), until the GPU is done %.3f ms"%(pipelined, hs[10], ts[10]))
    t0=time.perf_counter()
    for _ in range(50): r(pts)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("   back to back: host %.3f ms per replay, GPU %.3f ms per replay"%((t1-t0)*20, (t2-t0)*20))
