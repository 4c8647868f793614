"""dev probe: forward deviation and speed of the dense tail in f32 / bf16 / f16 (eval mode, KITTI cfg 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch
import bench

dev = torch.device("cuda:0")
cfg, ds, model, opt, sched = bench.build(2, dev, "f32")
model.eval()
batch = bench.make_batches(ds, 2, 4, 0, dev, n=1)[0]
# randomise BN stats a bit so activations are not degenerate
g = torch.Generator(device="cpu").manual_seed(1)
for m in model.modules():
    if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
        m.running_var.copy_((torch.rand(m.num_features, generator=g) * 0.5 + 0.25).to(dev))
        m.running_mean.copy_((torch.randn(m.num_features, generator=g) * 0.05).to(dev))

def run(dtype):
    bd = dict(batch)
    with torch.no_grad():
        for m in model.module_list[:3]:
            bd = m(bd)
        if dtype is None:
            bd = model.backbone_2d(bd)
        else:
            with torch.autocast("cuda", dtype=dtype):
                bd = model.backbone_2d(bd)
        bd = model.dense_head(bd)
    return bd["batch_box_preds"].float(), bd["batch_cls_preds"].float()

def timeit(dtype, n=10):
    for _ in range(3): run(dtype)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): run(dtype)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

b32, c32 = run(None)
print("f32 fwd %.2f ms / 4 frames" % (timeit(None) * 1e3))
for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    b, c = run(dt)
    print("%s fwd %.2f ms | max|dbox| %.3e  mean|dbox| %.3e | max|dlogit| %.3e  (logit range %.2f)" % (
        name, timeit(dt) * 1e3, float((b - b32).abs().max()), float((b - b32).abs().mean()),
        float((c - c32).abs().max()), float(c32.abs().max())))
