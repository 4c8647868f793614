"""Stability check (GPU box): 400 training steps of BASELINE configs[1] through bench.py's own Step — the loss stays finite,
the device status word stays clean and allocated / reserved memory stay flat (the side-stream weight gradients keep their
inputs alive until the end-of-backward join; with record_stream() instead the allocator's reserve grew to 18 GB).
usage: python tools/longrun_check.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tsm-det-pointcloud-_amd"))
import torch
import bench
sys.argv = ["bench.py"]
# reuse bench's builder: build(cfg_id, device, dense_dtype)
dev = torch.device("cuda:0")
from pcdet_amd.utils.miopen_db import use_tuned_db
use_tuned_db()
cfg, ds, model, opt, sched = bench.build(2, dev, "f32")
batches = bench.make_batches(ds, 2, 4, 0, dev, n=4)
from pcdet_amd.models.inference import static_caps_for
caps = static_caps_for(model, 4, max(int(b["points"].shape[0]) for b in batches), training=True)
step = bench.Step(model, opt, sched, cfg.OPTIMIZATION.GRAD_NORM_CLIP, "train", "f32", static_caps=caps)
model.train()
losses = []
for i in range(400):
    loss = step(batches[i % 4])
    if i % 50 == 0:
        torch.cuda.synchronize()
        losses.append(float(loss.detach()))
        print(i, "loss %.4f" % losses[-1], "allocated %.0f MB reserved %.0f MB" % (torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6), flush=True)
from spx import ops
ops.check_status(dev)
print("finite:", all(l == l and abs(l) < 1e6 for l in losses))
