"""Dev probe (GPU box): how much of a training step's wall time the HOST spends issuing it (process CPU time of the main
thread between two synchronisations), static-capacity path.  If this approaches the step time the step is host-bound."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch

import bench

dev = torch.device("cuda:0")
from pcdet_amd.utils.miopen_db import use_tuned_db
use_tuned_db()
cfg, ds, model, opt, sched = bench.build(2, dev, "f32")
model.train()
batches = bench.make_batches(ds, 2, 4, 0, dev)
from pcdet_amd.models.inference import static_caps_for
caps = static_caps_for(model, 4, max(int(b["points"].shape[0]) for b in batches), training=True)
step = bench.Step(model, opt, sched, cfg.OPTIMIZATION.GRAD_NORM_CLIP, "train", "f32", static_caps=caps)
for i in range(8):
    step(batches[i % 2])
torch.cuda.synchronize()
n = 40
w0, c0 = time.perf_counter(), time.thread_time()
t_issue = 0.0
for i in range(n):
    a = time.perf_counter()
    step(batches[i % 2])
    t_issue += time.perf_counter() - a
torch.cuda.synchronize()
w1, c1 = time.perf_counter(), time.thread_time()
print("step wall %.2f ms ; host issue (wall inside step(), no sync) %.2f ms ; main-thread CPU %.2f ms per step"
      % ((w1 - w0) / n * 1e3, t_issue / n * 1e3, (c1 - c0) / n * 1e3))
