"""dev probe: which python call sites launch the small torch kernels of a training step (torch.profiler with stacks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda:0")
cfg, ds, model, opt, sched = bench.build(2, dev, "f32")
model.train()
batches = bench.make_batches(ds, 2, 4, 0, dev)
step = bench.Step(model, opt, sched, 10.0, "train", "f32")
for i in range(3):
    step(batches[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(2):
        step(batches[i % 2])
    torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::copy_", "aten::cat", "aten::add_", "aten::fill_", "aten::clamp", "aten::mul", "aten::add", "aten::zero_"]
rows = prof.key_averages(group_by_stack_n=6)
for r in sorted(rows, key=lambda r: -r.device_time_total):
    if r.key in want and r.device_time_total > 20:
        stack = [s for s in r.stack if "site-packages/torch" not in s and "<built-in" not in s][:4]
        print("%-16s n=%3d cuda %8.1f us | %s" % (r.key, r.count, r.device_time_total / 2.0, " <- ".join(s.split("/")[-1] for s in stack)))
