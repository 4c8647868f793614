"""Dev probe (GPU box): does torch.ops.aten.miopen_convolution_relu (conv + bias + ReLU through MIOpen's fusion API) beat
conv2d followed by one libspx bias/affine + ReLU pass on the BEV backbone's shapes (fp32, channels_last)?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F

from spx import ops

dev = torch.device("cuda:0")


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


for c, h, w in ((128, 200, 176), (256, 100, 88)):
    x = torch.randn(4, c, h, w, device=dev).to(memory_format=torch.channels_last)
    wt = (torch.randn(c, c, 3, 3, device=dev) * 0.03).to(memory_format=torch.channels_last)
    b = torch.randn(c, device=dev)
    one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    with torch.no_grad():
        t_conv = t(lambda: F.conv2d(x, wt, None, 1, 1))
        y = F.conv2d(x, wt, None, 1, 1)
        rows = y.permute(0, 2, 3, 1).reshape(-1, c)
        t_bn = t(lambda: ops.bn_apply(rows, zero, one, one, b, True))
        try:
            f = lambda: torch.ops.aten.miopen_convolution_relu(x, wt, b, [1, 1], [1, 1], [1, 1], 1)
            z = f()
            ref = torch.relu(y + b[None, :, None, None])
            err = float((z - ref).abs().max())
            t_f = t(f)
            print("C=%d %dx%d: conv2d %.1f us + libspx bias+relu pass %.1f us = %.1f | miopen_convolution_relu %.1f us (max|diff| %.2e)"
                  % (c, h, w, t_conv, t_bn, t_conv + t_bn, t_f, err))
        except Exception as e:
            print("C=%d: miopen_convolution_relu failed: %s" % (c, e))
