"""dev probe: where a training step spends GPU time, by section (HIP events, KITTI cfg 2 batch 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch
import bench

dev = torch.device("cuda:0")
cfg, ds, model, opt, sched = bench.build(2, dev, "f32")
model.train()
batches = bench.make_batches(ds, 2, 4, 0, dev)
step = bench.Step(model, opt, sched, 10.0, "train", "f32")
for i in range(3):
    step(batches[i % 2])
torch.cuda.synchronize()
acc = {}
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
N = 5
for it in range(N):
    marks = [("start", ev())]
    bd = dict(batches[it % 2])
    opt.zero_grad(set_to_none=True)
    for name, m in zip(["vfe+voxelize", "backbone_3d", "map_to_bev", "backbone_2d"], model.module_list[:4]):
        bd = m(bd); marks.append((name, ev()))
    head = model.dense_head
    x = bd["spatial_features_2d"]
    cls_preds = head.conv_cls(x).permute(0, 2, 3, 1).contiguous()
    box_preds = head.conv_box(x).permute(0, 2, 3, 1).contiguous()
    dir_preds = head.conv_dir_cls(x).permute(0, 2, 3, 1).contiguous()
    head.forward_ret_dict.update(cls_preds=cls_preds, box_preds=box_preds, dir_cls_preds=dir_preds)
    marks.append(("head convs", ev()))
    head.forward_ret_dict.update(head.assign_targets(gt_boxes=bd["gt_boxes"])); marks.append(("assign_targets", ev()))
    loss, tb = head.get_loss(); marks.append(("losses", ev()))
    loss.backward(); marks.append(("backward (all)", ev()))
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0, foreach=True); marks.append(("clip_grad_norm", ev()))
    opt.step(); marks.append(("optimizer", ev()))
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = sum(acc.values())
for k, v in acc.items():
    print("%-18s %7.3f ms" % (k, v / N))
print("%-18s %7.3f ms" % ("TOTAL", tot / N))
