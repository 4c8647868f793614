"""Dev tool (GPU box): is the eager training step of bench.py bound by the host (launch issue) or by the GPU?  Times, per
step, how long the host needs to ISSUE the step (step() returning) and how long the GPU needs to finish it, at steady state.

python tools/cpu_bound_probe.py [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cfg", type=int, default=2)
    a = ap.parse_args()
    import bench
    from pcdet_amd.datasets import synthetic
    from pcdet_amd.models.inference import static_caps_for
    from pcdet_amd.utils.miopen_db import use_tuned_db
    use_tuned_db()
    dev = torch.device("cuda:0")
    batch = synthetic.CONFIGS[a.cfg]["batch"]
    cfg, ds, model, opt, sched = bench.build(a.cfg, dev, "f32")
    model.train()
    batches = bench.make_batches(ds, a.cfg, batch, 0, dev)
    caps = static_caps_for(model, batch, max(int(b["points"].shape[0]) for b in batches), training=True)
    step = bench.Step(model, opt, sched, cfg.OPTIMIZATION.GRAD_NORM_CLIP, "train", "f32", static_caps=caps)
    for i in range(12):
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    # how long the main stream waits for the weight-gradient stream at the end of the backward pass
    from spx import functional as F_
    waits = []

    def join_timed(main, side, keep):
        def _join():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            main.wait_stream(side)
            e1.record(main)
            waits.append((e0, e1))
            keep.clear()
            F_._DEFERRED.clear()
        torch.autograd.Variable._execution_engine.queue_callback(_join)
    orig = F_._join_after_backward
    F_._join_after_backward = join_timed
    per_step = []
    for i in range(10):
        del waits[:]
        step(batches[i % len(batches)])
        torch.cuda.synchronize()
        per_step.append(sum(a.elapsed_time(b) for a, b in waits))
    F_._join_after_backward = orig
    per_step.sort()
    print("end-of-backward join: main stream waits %.3f ms (median of 10 steps; min %.3f max %.3f), %d deferred gradients" % (
        per_step[5], per_step[0], per_step[-1], len(waits)))
    issue, total = [], []
    for i in range(a.steps):
        t0 = time.perf_counter()
        step(batches[i % len(batches)])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    issue.sort(), total.sort()
    print("one step at a time (sync after each): host issue median %.2f ms, step complete median %.2f ms" % (
        issue[len(issue) // 2], total[len(total) // 2]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(batches[i % len(batches)])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("back to back: host issued %d steps in %.2f ms/step, GPU done after %.2f ms/step" % (
        a.steps, (t1 - t0) * 1e3 / a.steps, (t2 - t0) * 1e3 / a.steps))


if __name__ == "__main__":
    main()
