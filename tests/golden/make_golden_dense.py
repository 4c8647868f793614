"""Generate tests/golden/dense_conv_*.npz — the independent known-answer vectors for rows a7–a9, a12, a13.

Expected outputs come from PyTorch's DENSE ops only (F.conv3d on the densified input and its autograd),
never from the oracle or the HIP library, so they pin both:
  * active OUTPUT SET of a regular sparse conv  = nonzero(conv3d(occupancy, ones))   (integer, exact)
  * values at active outputs                    = conv3d(dense_in, W.permute(0,4,1,2,3))  (fp32 round-off)
  * submanifold conv                            = the same dense conv sampled at the INPUT sites
  * dgrad / wgrad                               = torch autograd through that dense expression
This is the check SURVEY.md §8(c) describes ("sparse conv == dense F.conv3d sampled at active outputs").

Run (build container or GPU box, CPU only):  python tests/golden/make_golden_dense.py
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))

# the four geometries of VoxelBackBone8x (reference spconv_backbone.py:86-122) + a dilated/odd one
CASES = {
    "subm_k3": dict(subm=True, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), cin=5, cout=7),
    "sp_k3s2p1": dict(subm=False, k=(3, 3, 3), s=(2, 2, 2), p=(1, 1, 1), cin=6, cout=4),
    "sp_k3s2p011": dict(subm=False, k=(3, 3, 3), s=(2, 2, 2), p=(0, 1, 1), cin=4, cout=5),
    "sp_k311s211p0": dict(subm=False, k=(3, 1, 1), s=(2, 1, 1), p=(0, 0, 0), cin=8, cout=3),
    "sp_k3s1p1": dict(subm=False, k=(3, 3, 3), s=(1, 1, 1), p=(1, 1, 1), cin=3, cout=4),
}


def make_case(name, cfg, seed, batch=2, shape=(9, 16, 12), n_active=300):
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    cells = batch * D * H * W
    lin = torch.randperm(cells, generator=g)[:n_active]  # deliberately UNSORTED row order
    b = lin // (D * H * W)
    r = lin % (D * H * W)
    idx = torch.stack([b, r // (H * W), (r // W) % H, r % W], 1).int()
    feat = torch.randn(n_active, cfg["cin"], generator=g)
    w = torch.randn(cfg["cout"], *cfg["k"], cfg["cin"], generator=g) * 0.2  # [Cout,kz,ky,kx,Cin]

    feat_t = feat.clone().requires_grad_(True)
    w_t = w.clone().requires_grad_(True)
    # channels-last scatter (autograd-friendly), then permute to NCDHW
    dense = torch.zeros(batch, D, H, W, cfg["cin"])
    dense = dense.index_put((idx[:, 0].long(), idx[:, 1].long(), idx[:, 2].long(), idx[:, 3].long()), feat_t)
    dense = dense.permute(0, 4, 1, 2, 3)
    occ = torch.zeros(batch, 1, D, H, W)
    occ[idx[:, 0].long(), 0, idx[:, 1].long(), idx[:, 2].long(), idx[:, 3].long()] = 1.0
    y = F.conv3d(dense, w_t.permute(0, 4, 1, 2, 3), stride=cfg["s"], padding=cfg["p"])
    if cfg["subm"]:
        out_idx = idx.clone()  # submanifold: outputs == inputs, same order
    else:
        cov = F.conv3d(occ, torch.ones(1, 1, *cfg["k"]), stride=cfg["s"], padding=cfg["p"])
        out_idx = (cov[:, 0] > 0.5).nonzero().int()  # ascending (b,z,y,x) == canonical order
    o = out_idx.long()
    out = y[o[:, 0], :, o[:, 1], o[:, 2], o[:, 3]]
    dout = torch.randn(out.shape, generator=g)
    (out * dout).sum().backward()
    return dict(idx=idx.numpy(), feat=feat.numpy(), w=w.numpy(), batch=np.int32(batch),
                shape=np.array(shape, np.int32), k=np.array(cfg["k"], np.int32), s=np.array(cfg["s"], np.int32),
                p=np.array(cfg["p"], np.int32), subm=np.int32(cfg["subm"]), out_idx=out_idx.numpy(),
                out=out.detach().numpy(), dout=dout.numpy(), dfeat=feat_t.grad.numpy(), dw=w_t.grad.numpy(),
                out_spatial=np.array(y.shape[2:], np.int32))


def main():
    for i, (name, cfg) in enumerate(CASES.items()):
        d = make_case(name, cfg, seed=100 + i)
        np.savez_compressed(os.path.join(HERE, "dense_conv_%s.npz" % name), **d)
        print(name, "n_in", d["idx"].shape[0], "n_out", d["out_idx"].shape[0])


if __name__ == "__main__":
    main()
