"""dev probe: host-side cost of one spx.ops call (python wrapper + ctypes + launches), measured on tiny inputs so that
the GPU never limits: the small layers at the end of the backward pass are bound by this, not by their kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsm-det-pointcloud-_amd"))
import torch
from spx import ops
import spx
from spx import functional as F_

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = 2000
idx = torch.stack([torch.zeros(n, dtype=torch.int32), torch.randint(0, 8, (n,), generator=g, dtype=torch.int32),
                   torch.randint(0, 64, (n,), generator=g, dtype=torch.int32), torch.randint(0, 64, (n,), generator=g, dtype=torch.int32)], 1)
idx = torch.unique(idx, dim=0).to(dev)
n = idx.shape[0]
rb = ops.subm_rulebook(idx, 1, [8, 64, 64], (3, 3, 3))
x = torch.randn(n, 32, device=dev)
w = torch.randn(32, 3, 3, 3, 32, device=dev)
wp = ops.pack_weight(w, 0)
bn = torch.nn.BatchNorm1d(32).to(dev).train()
dy = torch.randn(n, 32, device=dev)


def host(name, fn, iters=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s host %7.1f us / call   (queue drained %.1f ms after the loop)" % (name, (t1 - t0) / iters * 1e6, (t2 - t1) * 1e3))


host("torch.empty_like(x)", lambda: torch.empty_like(x))
host("ops.subm_rulebook", lambda: ops.subm_rulebook(idx, 1, [8, 64, 64], (3, 3, 3)))
host("ops.conv_gemm 32->32", lambda: ops.conv_gemm(x, wp, 32, 27, rb.pair, rb.ld, rb.n_out))
host("ops.conv_wgrad 32->32", lambda: ops.conv_wgrad(x, dy, rb.pair, rb.ld, rb.n_out, tuple(w.shape)))
host("ops.pack_weight", lambda: ops.pack_weight(w, 0))
host("ops.bn_relu_fwd", lambda: ops.bn_relu_fwd(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.01, 1e-3, True))
y, mean, invstd = ops.bn_relu_fwd(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.01, 1e-3, True)
host("ops.bn_relu_bwd", lambda: ops.bn_relu_bwd(x, dy, bn.weight, bn.bias, mean, invstd, True))
conv = spx.SubMConv3d(32, 32, 3, bias=False, indice_key="k").to(dev)
seq = spx.SparseSequential(conv, bn, torch.nn.ReLU())
st = spx.SparseConvTensor(x, idx, [8, 64, 64], 1)
st.indice_dict["k"] = rb


def layer():
    xx = x.detach().requires_grad_(True)
    t = spx.SparseConvTensor(xx, idx, [8, 64, 64], 1)
    t.indice_dict["k"] = rb
    out = seq(t)
    out.features.backward(dy)


host("SubMConv3d+BN+ReLU layer fwd+bwd (autograd)", layer, 100)

if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(200):
        layer()
    pr.disable()
    torch.cuda.synchronize()
    st_ = pstats.Stats(pr)
    st_.sort_stats("tottime").print_stats(28)
