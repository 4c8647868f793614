"""Dev probe (GPU box): the round-1 graph-replay stall, looked for INSIDE the library.  Builds a variant of libspx whose
workspace clears are hipMemsetAsync calls again (-DSPX_FILL_USE_MEMSET) into gpurun_out/, captures the sync-free backbone
forward (voxelise -> 8 rule tables -> 12 convs -> densify) into a hipGraph with each build and replays it 4 times on two
different batches, checking after every replay the voxel count, the BEV map against the eager result (relative difference: the capacity path may pick another kernel schedule) and the device status
word (SPX_ERR_TABLE_FULL = a probe sequence met a table that was not cleared).  Probe loops are bounded since round 2, so a
stale workspace shows up as an error code / a mismatch here, not as a hang.

    python tools/graph_memset_in_lib.py            # prints one line per (build, replay)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd")):
    sys.path.insert(0, p)


def build_variant():
    csrc = os.path.join(ROOT, "tsm-det-pointcloud-_amd", "csrc")
    out = os.path.join(ROOT, "gpurun_out", "memset_variant")
    os.makedirs(out, exist_ok=True)
    objs = []
    for f in sorted(x for x in os.listdir(csrc) if x.endswith(".hip")):
        o = os.path.join(out, f[:-4] + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
                               "-DSPX_FILL_USE_MEMSET", "-c", os.path.join(csrc, f), "-o", o])
        objs.append(o)
    lib = os.path.join(out, "libspx_memset.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


def run(lib_path, tag):
    code = r'''
import os, sys
sys.path[:0] = [%r, %r]
import torch
from spx import _lib
if %r:
    _lib.LIB_PATH = %r
from spx import ops
from pcdet_amd.config import AttrDict, cfg_from_yaml_file
from pcdet_amd.datasets import SyntheticDataset, synthetic
from pcdet_amd.models import build_network
from pcdet_amd.models.inference import GraphedDetector
cfg = cfg_from_yaml_file(os.path.join(%r, "tsm-det-pointcloud-_amd/tools/cfgs/kitti_models/second.yaml"), AttrDict())
ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, False, cfg_id=2)
torch.manual_seed(0)
dev = torch.device("cuda:0")
model = build_network(cfg.MODEL, 3, ds).to(dev).eval()
pts = [torch.from_numpy(synthetic.make_batch(2, 2, start_frame=2 * i)["points"]).to(dev) for i in range(2)]
def eager(p):
    bd = {"points": p, "batch_size": 2}
    with torch.no_grad():
        for m in model.module_list[:3]:
            bd = m(bd)
    return bd["voxel_coords"].shape[0], bd["spatial_features"].clone()
refs = [eager(p) for p in pts]
g = GraphedDetector(model, 2, int(max(p.shape[0] for p in pts) * 1.05) + 64, n_modules=3)
for rep in range(4):
    p = pts[rep %% 2]
    out = g(p)
    torch.cuda.synchronize()
    nv = int(out["counts"]["voxels"])
    ref = refs[rep %% 2][1]
    same = "%%.1e" %% float((out["spatial_features"] - ref).abs().max() / ref.abs().max())
    st = int(ops.status_word(dev).item())
    ops.status_word(dev).zero_()
    print("%s replay %%d: voxels %%d (eager %%d)  BEV map max rel diff vs eager: %%s  device status %%d" %% (rep + 1, nv, refs[rep %% 2][0], same, st))
''' % (ROOT, os.path.join(ROOT, "tsm-det-pointcloud-_amd"), bool(lib_path), lib_path, ROOT, tag)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    if r.returncode != 0:
        sys.stdout.write("%s FAILED rc %d\n%s\n" % (tag, r.returncode, r.stderr[-2000:]))


if __name__ == "__main__":
    run(None, "fill-kernel build (shipped)")
    run(build_variant(), "hipMemsetAsync build      ")
