"""Dev tool (GPU box): fp32 error of Winograd F(4x4, 3x3) against float64 at the BEV shapes of BASELINE configs[1] / [2],
beside F(2x2, 3x3) (the shipped kernel, and the same arithmetic emulated in torch) — VERDICT round 2, item 6: "settle
F(4x4, 3x3) with a measurement".  The F(4x4) arithmetic is EMULATED: fp32 transforms with the Lavin & Gray matrices and one
fp32 batched GEMM per transform position (rocBLAS sgemm on gfx950 is exact fp32 — there is no TF32), i.e. the error a HIP
kernel of that decomposition would have up to the summation order inside the 36 GEMMs.  Bar of the shipped kernels:
|err| <= 2e-5 * max(1, max|ref|) (tests/test_gpu_wino.py).

python tools/wino_f4_error.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)


def mats(m):
    if m == 2:
        bt = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        g = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
        at = [[1, 1, 1, 0], [0, 1, -1, -1]]
    else:
        bt = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
              [0, 4, 0, -5, 0, 1]]
        g = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
             [0, 0, 1]]
        at = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    return [torch.tensor(v, dtype=torch.float64) for v in (bt, g, at)]


def wino_emulated(x, w, m):
    """conv2d(x, w, padding=1) through F(m x m, 3 x 3) with every product and sum in fp32."""
    dev = x.device
    bt, g, at = [t.to(dev).float() for t in mats(m)]
    n, c, h, wd = x.shape
    co = w.shape[0]
    a = m + 2
    ty, tx = (h + m - 1) // m, (wd + m - 1) // m
    xp = F.pad(x, (1, tx * m + 1 - wd, 1, ty * m + 1 - h))
    patches = xp.unfold(2, a, m).unfold(3, a, m)                      # [n, c, ty, tx, a, a]
    v = torch.einsum("ia,nctuab,jb->ijntuc", bt, patches, bt)          # B^T d B, fp32
    u = torch.einsum("ia,ocab,jb->ijco", g, w, g)                      # G g G^T
    mm = torch.matmul(v.reshape(a * a, n * ty * tx, c), u.reshape(a * a, c, co))       # 16 / 36 fp32 GEMMs
    mm = mm.reshape(a, a, n, ty, tx, co)
    y = torch.einsum("pi,ijntuo,qj->notpuq", at, mm, at)               # A^T M A -> [n, co, ty, m, tx, m]
    return y.reshape(n, co, ty * m, tx * m)[:, :, :h, :wd]


def conv64(x, w):
    n, c, h, wd = x.shape
    xp = F.pad(x.double().permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    y = None
    for a in range(3):
        for b in range(3):
            part = xp[:, a:a + h, b:b + wd, :].reshape(n * h * wd, c) @ w.double()[:, :, a, b].t()
            y = part if y is None else y + part
    return y.reshape(n, h, wd, -1).permute(0, 3, 1, 2)


def main():
    from spx import ops
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    print("%-22s %12s %12s %12s %12s   (max|err| / max(1, max|ref|); bar 2e-5)" % (
        "shape", "F(2x2) kernel", "F(2x2) emul.", "F(4x4) emul.", "vendor direct"))
    for n, c, h, wd in ((4, 128, 200, 176), (4, 256, 100, 88), (2, 128, 188, 188), (2, 256, 94, 94)):
        worst = [0.0, 0.0, 0.0, 0.0]
        for seed, scale_in in ((0, 1.0), (1, 1.0), (2, 4.0)):          # the third: activations with a larger spread
            g = torch.Generator().manual_seed(seed)
            x = (torch.randn((n, c, h, wd), generator=g) * scale_in).to(dev).contiguous(memory_format=torch.channels_last)
            if seed == 2:
                x = torch.relu(x)                                       # post-ReLU statistics (what the layers see)
            w = (torch.randn((c, c, 3, 3), generator=g) / np.sqrt(9 * c)).to(dev)
            ref = conv64(x, w)
            den = max(1.0, float(ref.abs().max()))
            outs = [ops.conv2d_wino(x, ops.wino_weight(w), c), wino_emulated(x, w, 2), wino_emulated(x, w, 4),
                    F.conv2d(x, w, padding=1)]
            for i, y in enumerate(outs):
                worst[i] = max(worst[i], float((y.double() - ref).abs().max()) / den)
            del ref, outs
        print("%-22s %12.2e %12.2e %12.2e %12.2e" % ("%dx%dx%dx%d" % (n, c, h, wd), *worst))


if __name__ == "__main__":
    main()
