"""Dev tool (GPU box): time one sparse-conv API call on a real layer of the cfg-2 batch with ONE source file of libspx
rebuilt with extra -D flags (ablations, A/B switches); the other objects are the shipped ones from csrc/build.

python tools/abl.py --file conv_wgrad.hip --defs SPX_WGL_NO_DMA --what wgrad --layer conv3.1.0"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", required=True)
    ap.add_argument("--defs", default="")
    ap.add_argument("--what", default="wgrad", choices=["wgrad", "fwd", "dgrad"])
    ap.add_argument("--layer", default="conv3.1.0")
    args = ap.parse_args()
    csrc = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "csrc")
    out = "/tmp/spx_abl"
    os.makedirs(out, exist_ok=True)
    o = os.path.join(out, args.file[:-4] + ".o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
                           *["-D" + x for x in args.defs.split(",") if x], "-c", os.path.join(csrc, args.file), "-o", o])
    objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build"))
            if f.endswith(".o") and f != args.file[:-4] + ".o"]
    lib_path = os.path.join(out, "libspx_abl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, o, *objs])
    import numpy as np
    import torch
    from spx import _lib
    _lib.LIB_PATH = lib_path
    from kbench import backbone8x_layers, timeit
    from pcdet_amd.datasets import synthetic
    from spx import ops
    dev = torch.device("cuda:0")
    spec = synthetic.CONFIGS[2]
    geom, batch = spec["geom"], spec["batch"]
    b = synthetic.make_batch(2, batch)
    pts = torch.from_numpy(b["points"]).to(dev)
    vox = ops.voxelize(pts, geom["point_cloud_range"], geom["voxel_size"], 5, geom["max_voxels"]["train"],
                       batch_size=batch, batch_col=0, xyz_col=1, feat_col=1, want_voxels=False)
    gs = synthetic.grid_size_of(geom)
    shape = [int(gs[2]) + 1, int(gs[1]), int(gs[0])]
    idx = vox["coords"]
    books = {}
    for name, cin, cout, ks, st, pd, ctype, key in backbone8x_layers(geom["num_point_features"]):
        if key not in books:
            books[key] = ops.subm_rulebook(idx, batch, shape, ks) if ctype == "subm" else \
                ops.conv_rulebook(idx, batch, shape, ks, st, pd)
        rb = books[key]
        if name == args.layer:
            break
        if ctype != "subm":
            idx, shape = rb.out_indices, rb.out_shape
    K = rb.kvol
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rb.n_in, cin, generator=g).to(dev)
    w = (torch.randn(cout, *ks, cin, generator=g) / np.sqrt(K * cin)).to(dev)
    dout = torch.randn(rb.n_out, cout, generator=g).to(dev)
    if args.what == "wgrad":
        counts = ops.wgrad_counts(rb.pair, rb.ld, K, rb.n_out)
        f = (lambda: ops.conv_wgrad(x, dout, rb.pair, rb.ld, rb.n_out, tuple(w.shape), counts=counts))
    elif args.what == "fwd":
        wp = ops.pack_weight(w, 0)
        f = (lambda: ops.conv_gemm(x, wp, cout, K, rb.pair, rb.ld, rb.n_out))
    else:
        wt = ops.pack_weight(w, 1)
        f = (lambda: ops.conv_gemm(dout, wt, cin, K, rb.pair, rb.ld, rb.n_in, flip_k=True))
    t = timeit(f, 20)
    print("layer %s  %s  defs [%s]  %.1f us" % (args.layer, args.what, args.defs, t * 1e6))


if __name__ == "__main__":
    main()
