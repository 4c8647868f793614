"""bench.py — headline benchmark: point-cloud frames/s (fwd+bwd+optimizer step) of the SECOND detector on the
VoxelBackBone8x hot path, KITTI-shaped synthetic frames (BASELINE.json configs[1]: ~20k points, 16k active voxels
per frame, batch 4 per GPU), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = PointToVoxel scatter + MeanVFE -> VoxelBackBone8x (12 sparse convs, 8 rulebooks) -> BEV collapse ->
BaseBEVBackbone -> AnchorHeadSingle -> target assignment + losses -> backward -> grad clip -> AdamW(one-cycle) step,
on points already resident in HBM.  Frames are sharded over ranks (weak scaling); the only collective is DDP's
gradient all-reduce (RCCL over xGMI).  Rank 0 prints ONE JSON line.

`--gpus N` without a torchrun environment: the parent starts the N ranks itself (torch.distributed.run, 127.0.0.1) BEFORE
it touches the GPU, forwards rank 0's line and exits with the children's code; under torchrun WORLD_SIZE must equal --gpus.

Besides the contract fields the line carries
  extras        (N = 1 only, each timed in-process after the headline) `forward` (cfg 2 full detector, eager and one
                hipGraph), `backbone_forward` (voxelise -> VoxelBackBone8x -> densify only, eager and hipGraph, with its own
                roofline from the per-layer algorithmic bytes / FLOPs of SURVEY.md §8(d)), `rulebook_mvoxels_s` (submanifold
                and strided tables), `cfg3_fwd_bwd` (Waymo 2 x 80k voxels, full training step) and `cfg5_forward` (Waymo
                5-frame concat, 300k voxels);
  roofline      for the dominant hand-written kernel, from HIP events recorded around every launch of it on the
                launch stream during extra instrumented steps of the same workload (the timed region itself runs
                un-instrumented), algorithmic FLOPs/bytes per SURVEY.md §8(d);
  cpu_baseline  the CPU oracle (numpy per-offset gather-GEMM-scatter restatement of spconv's CPU algorithm, all
                host cores, + torch-CPU dense tail) timed on a bounded sample: one frame of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "tsm-det-pointcloud-_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12      # B/s   (MI355X_MICROARCH.md: HBM3E 8 TB/s spec)
MFMA_F32_PEAK = 157.3e12  # FLOP/s (dense fp32 MFMA, v_mfma_f32_16x16x4_f32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cfg", type=int, default=2, help="synthetic config id (BASELINE.md §4); 2 = KITTI batch 4")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU (default: the config's)")
    ap.add_argument("--mode", default="train", choices=["train", "fwd", "fwd-graph"],
                    help="fwd-graph: sync-free forward captured in one hipGraph (pcdet_amd/models/inference.py)")
    ap.add_argument("--dense-dtype", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--dense-layout", default="nhwc", choices=["nhwc", "nchw"], help="memory format of the BEV backbone")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark = True (MIOpen find mode)")
    ap.add_argument("--no-miopen-db", action="store_true",
                    help="do not load the tuned MIOpen user database shipped in tsm-det-pointcloud-_amd/miopen_db (solver "
                         "choices of MIOpen's own tuner for the dense tail's convolutions; immediate mode then uses MIOpen's "
                         "system database)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward / backbone / rulebook / Waymo extras")
    ap.add_argument("--graph", action="store_true",
                    help="single-GPU training: replay forward + backward as one hipGraph (GraphedTrainStep) instead of issuing "
                         "them kernel by kernel; measured SLOWER than eager on MI355X (298 vs 310 frames/s, DESIGN.md 6b)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="rehearse the N > 1 code path with one rank: RCCL process group of size 1 + the DDP wrapper")
    ap.add_argument("--plain-ddp", action="store_true",
                    help="N > 1: torch's DistributedDataParallel as the reference wraps it, without the late reduction of the "
                         "convolution weights (A/B)")
    ap.add_argument("--dynamic", action="store_true",
                    help="training with exact-size sparse tensors (one host read per strided rule table) instead of the "
                         "host-sync-free static-capacity path")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    return ap.parse_args()


DENSE_LAYOUT = "nhwc"


def build(cfg_id, device, dense_dtype):
    from pcdet_amd.config import AttrDict, cfg_from_yaml_file
    from pcdet_amd.datasets import SyntheticDataset, synthetic
    from pcdet_amd.models import build_network
    from tools.train_utils.optimization import build_optimizer, build_scheduler
    name = synthetic.CONFIGS[cfg_id]["geom"]["name"]
    ymodel = "waymo_models/second.yaml" if name == "waymo" else "kitti_models/second.yaml"
    cfg = cfg_from_yaml_file(os.path.join(PKG, "tools", "cfgs", ymodel), AttrDict())
    ds = SyntheticDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, training=True, cfg_id=cfg_id)
    cfg.MODEL.VFE.VOXELIZE.MAX_NUMBER_OF_VOXELS = ds.max_voxels
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    model.to(device)
    if device.type == "cuda" and DENSE_LAYOUT == "nhwc":
        model.backbone_2d.to(memory_format=torch.channels_last)
    if device.type == "cuda":
        model.map_to_bev_module.channels_last = DENSE_LAYOUT == "nhwc"
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(optimizer, 1000, 80, -1, cfg.OPTIMIZATION)
    return cfg, ds, model, optimizer, sched


def make_batches(ds, cfg_id, batch, rank, device, n=2):
    from pcdet_amd.datasets import synthetic
    out = []
    for j in range(n):
        b = synthetic.make_batch(cfg_id, batch, start_frame=(rank * n + j) * batch)
        out.append({"points": torch.from_numpy(b["points"]).to(device), "gt_boxes": torch.from_numpy(b["gt_boxes"]).to(device),
                    "batch_size": batch})
    return out


class Step(object):
    def __init__(self, model, optimizer, sched, clip, mode, dense_dtype, static_caps=None):
        self.model, self.opt, self.sched, self.clip, self.mode = model, optimizer, sched, clip, mode
        self.static_caps = static_caps       # training at static row capacities: no host read anywhere in the step
        self.dense_dtype = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}[dense_dtype]
        self.it = 0
        core = model.module if hasattr(model, "module") else model
        if self.dense_dtype is not None:
            fwd = core.backbone_2d.forward
            dt = self.dense_dtype

            def amp_forward(data_dict):
                with torch.autocast("cuda", dtype=dt):
                    return fwd(data_dict)
            core.backbone_2d.forward = amp_forward

    def __call__(self, batch):
        bd = dict(batch)
        if self.mode == "fwd-graph":
            if getattr(self, "runner", None) is None:
                from pcdet_amd.models.inference import GraphedDetector
                core = self.model.module if hasattr(self.model, "module") else self.model
                self.runner = GraphedDetector(core, batch["batch_size"], int(batch["points"].shape[0] * 1.05) + 64)
            return self.runner(batch["points"])["batch_box_preds"]
        if self.mode == "fwd":
            with torch.no_grad():
                for m in (self.model.module if hasattr(self.model, "module") else self.model).module_list:
                    bd = m(bd)
            return bd["batch_box_preds"]
        self.sched.step(self.it)
        if getattr(self, "graphed", None) is not None:
            # forward + backward = one hipGraph replay (pcdet_amd/models/inference.py:GraphedTrainStep); gradients are
            # rewritten in place by the replay, so no zero_grad
            loss = self.graphed(batch["points"], batch["gt_boxes"])["loss"]
        else:
            self.opt.zero_grad(set_to_none=True)
            if self.static_caps is not None:
                bd["static_caps"] = self.static_caps
            ret, _tb, _ = self.model(bd)
            loss = ret["loss"].mean()
            loss.backward()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip, foreach=True)
        self.opt.step()
        self.it += 1
        return loss


# ------------------------------------------------------------------------------------------ instrumentation

class KernelTimer(object):
    """HIP events (torch.cuda.Event on the launch stream = torch's current stream, which is the stream every spx
    kernel is enqueued on) around each spx.ops call; algorithmic FLOPs / bytes per SURVEY.md §8(d)."""

    def __init__(self):
        self.rec = []      # (family, flops, bytes, ev0, ev1)
        self._saved = {}
        self._pairs = {}   # id(pair tensor) -> number of valid pairs P

    def _P(self, pair, n):
        key = (pair.data_ptr(), int(n))
        if key not in self._pairs:
            self._pairs[key] = int((pair[:, :n] >= 0).sum().item())
        return self._pairs[key]

    def _S(self, pair, n):
        """Live source rows of a rule table = 1 + the largest row it refers to (every live row of the layers measured here is
        some destination's neighbour); used where the caller runs at static capacity and src.shape[0] is the capacity."""
        key = ("s", pair.data_ptr(), int(n))
        if key not in self._pairs:
            self._pairs[key] = int(pair[:, :n].max().item()) + 1 if n > 0 else 0
        return self._pairs[key]

    def _conv_work(self, src, c_dst, kvol, pair, n_dst, d_n, d_n_src=None):
        """(flops, bytes) of one gather-GEMM per SURVEY.md section 8(d), as callables over the LIVE row counts."""
        cs = src.shape[1]

        def flops():
            return 2.0 * self._P(pair, self._live(d_n, n_dst)) * cs * c_dst

        def nbytes():
            nd = self._live(d_n, n_dst)
            ns = src.shape[0] if d_n is None else min(src.shape[0], self._S(pair, nd))   # static capacity: live source rows
            return 4.0 * (ns * cs + nd * c_dst + kvol * cs * c_dst + kvol * nd)
        return flops, nbytes

    def _timed(self, family, flops, nbytes, fn, *a, **k):
        """flops / nbytes: numbers, or callables evaluated after the launch has been timed (they may read device-side row
        counts: a host sync between the two events would be timed as kernel time)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        self.rec.append((family, flops, nbytes, e0, e1, out))
        return out

    @staticmethod
    def _live(d_n, n):
        return n if d_n is None else min(int(d_n.item()), int(n))

    def install(self):
        from spx import ops
        t = self
        names = ["conv_gemm", "conv_gemm_balanced", "conv_gemm_ring", "conv_ring_plan", "conv_plan", "conv_group", "conv_wgrad", "subm_rulebook", "conv_rulebook", "voxelize",
                 "densify", "densify_bwd", "pack_weight", "conv2d_wino", "conv2d_wino_wgrad", "wino_weight"]
        self._saved = {n: getattr(ops, n) for n in names}
        sv = self._saved

        def conv_gemm(src, w_packed, c_dst, kvol, pair, ld, n_dst, flip_k=False, scale=None, shift=None, relu=False,
                      d_n_dst=None):
            cs = src.shape[1]
            flops, nbytes = t._conv_work(src, c_dst, kvol, pair, n_dst, d_n_dst)
            # one family per template instantiation, named so that it maps 1:1 to a rocprofv3 row (k_conv_mfma<cs, cd, ..>)
            fam = "conv_gemm[mfma %dx%d]" % (cs, c_dst) if (cs % 16 == 0 and c_dst % 16 == 0) else "conv_gemm[valu]"
            return t._timed(fam, flops, nbytes, sv["conv_gemm"], src, w_packed, c_dst, kvol, pair, ld, n_dst, flip_k,
                            scale, shift, relu, d_n_dst)

        def conv_gemm_balanced(src, w_packed, c_dst, kvol, pair, ld, n_dst, plan, flip_k=False, scale=None, shift=None,
                               relu=False, d_n_dst=None, perm=None):
            cs = src.shape[1]
            flops, nbytes = t._conv_work(src, c_dst, kvol, pair, n_dst, d_n_dst)
            return t._timed("conv_gemm[mfma %dx%d balanced]" % (cs, c_dst), flops, nbytes, sv["conv_gemm_balanced"], src, w_packed,
                            c_dst, kvol, pair, ld, n_dst, plan, flip_k, scale, shift, relu, d_n_dst, perm)

        def conv_gemm_ring(src, w_packed, c_dst, kvol, pair, ld, n_dst, plan, flip_k=False, scale=None, shift=None,
                           relu=False, d_n_dst=None, perm=None, want_stats=False):
            cs = src.shape[1]
            flops, nbytes = t._conv_work(src, c_dst, kvol, pair, n_dst, d_n_dst)
            # family name = the rocprofv3 row k_conv_ring<cs, cd> (csrc/conv_ring.hip)
            return t._timed("conv_gemm[mfma %dx%d ring]" % (cs, c_dst), flops, nbytes, sv["conv_gemm_ring"], src, w_packed,
                            c_dst, kvol, pair, ld, n_dst, plan, flip_k, scale, shift, relu, d_n_dst, perm, want_stats)

        def conv_ring_plan(pair, ld, kvol, n_dst, d_n_dst=None):
            return t._timed("conv_ring_plan", 0.0, lambda: 4.0 * kvol * t._live(d_n_dst, n_dst), sv["conv_ring_plan"], pair, ld,
                            kvol, n_dst, d_n_dst)

        def conv_plan(pair, ld, kvol, n_dst, d_n_dst=None):
            return t._timed("conv_plan", 0.0, lambda: 4.0 * kvol * t._live(d_n_dst, n_dst), sv["conv_plan"], pair, ld, kvol, n_dst,
                            d_n_dst)

        def conv_group(pair, ld, kvol, n_dst, d_n_dst=None):
            # reads the table, writes perm and the grouped table
            return t._timed("conv_group", 0.0, lambda: 4.0 * (2 * kvol + 1) * t._live(d_n_dst, n_dst), sv["conv_group"], pair, ld,
                            kvol, n_dst, d_n_dst)

        def conv_wgrad(feat_in, dout, pair, ld, n_out, wshape, d_n_out=None, counts=None):
            cout, cin = wshape[0], wshape[-1]
            K = int(np.prod(wshape[1:-1]))

            def flops():
                return 2.0 * t._P(pair, t._live(d_n_out, n_out)) * cin * cout

            def nbytes():
                no = t._live(d_n_out, n_out)
                return 4.0 * (feat_in.shape[0] * cin + no * cout + K * no + K * cin * cout)
            # one family per template instantiation (k_wgrad_mfma<cin/16, cout/16, ..>), like the forward kernels
            return t._timed("conv_wgrad[mfma %dx%d]" % (cin, cout), flops, nbytes, sv["conv_wgrad"], feat_in, dout, pair, ld,
                            n_out, wshape, d_n_out, counts)

        def subm_rulebook(indices, batch_size, spatial_shape, ksize, dilation=(1, 1, 1), want_cnt=False, d_n=None, **kw):
            n = indices.shape[0]
            K = int(np.prod(ksize))
            return t._timed("subm_rulebook", 0.0, lambda: t._live(d_n, n) * (16.0 + K * 4.0), sv["subm_rulebook"], indices,
                            batch_size, spatial_shape, ksize, dilation, want_cnt, d_n, **kw)

        def conv_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation=(1, 1, 1),
                          want_cnt=False, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rb = sv["conv_rulebook"](indices, batch_size, spatial_shape, ksize, stride, padding, dilation, want_cnt, **kw)
            e1.record()
            n, K = indices.shape[0], int(np.prod(ksize))

            def nbytes():          # live rows on both sides (static capacities: counts stay on the device until now)
                ni = t._live(kw.get("d_n_in"), n)
                no = t._live(getattr(rb, "d_n_out", None), rb.n_out)
                extra = 0.0
                if getattr(rb, "subm_next", None) is not None:      # the level build also writes the output level's subm table
                    extra = no * 4.0 * int(np.prod(kw.get("subm_ksize") or (3, 3, 3)))
                return ni * 16.0 + K * no * 4.0 + K * ni * 4.0 + no * 16.0 + extra
            t.rec.append(("conv_rulebook", 0.0, nbytes, e0, e1, None))
            return rb

        def voxelize(points, *a, **k):
            t._pairs.clear()      # a new step: the allocator may hand a freed table's address to another table of the same size
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = sv["voxelize"](points, *a, **k)
            e1.record()
            c = out["mean"].shape[1] if out["mean"] is not None else 4

            def nbytes():
                nv = out["num_voxels"] if out["num_voxels"] is not None else int(out["d_num_voxels"].item())
                return 4.0 * points.shape[0] * c + 4.0 * nv * (c + 4)
            t.rec.append(("voxelize+meanvfe", 0.0, nbytes, e0, e1, None))
            return out

        def densify(features, indices, batch_size, spatial_shape, channels_last=False, d_n=None):
            n, c = features.shape
            cells = batch_size * int(np.prod(spatial_shape))
            return t._timed("densify(+memset)", 0.0, lambda: 4.0 * (t._live(d_n, n) * c + c * cells), sv["densify"], features,
                            indices, batch_size, spatial_shape, channels_last, d_n)

        def densify_bwd(ddense, indices, batch_size, spatial_shape, channels_last=False, d_n=None):
            n, c = indices.shape[0], ddense.shape[1]
            return t._timed("densify_bwd", 0.0, lambda: 8.0 * t._live(d_n, n) * c, sv["densify_bwd"], ddense, indices, batch_size,
                            spatial_shape, channels_last, d_n)

        def pack_weight(weight, mode, **kw):
            return t._timed("pack_weight", 0.0, 8.0 * weight.numel(), sv["pack_weight"], weight, mode, **kw)

        def conv2d_wino(x, u, cout, scale=None, shift=None, relu=False, out=None, stats=False):
            # dense 3x3 conv of the BEV backbone, Winograd F(2x2, 3x3): the kernel EXECUTES 16 multiply-adds per 2x2 output tile
            # and (ci, co) — 4 per output pixel — where the direct form has 9; `flops` is what runs on the MFMA, the
            # direct-form figure is 2.25x that (reported beside it).  bytes: map in + map out + the weight image.
            n, cin, h, w = x.shape
            tiles = n * ((h + 1) // 2) * ((w + 1) // 2)
            flops = 2.0 * 16 * tiles * cin * cout
            nbytes = 4.0 * (n * h * w * (cin + cout) + 16 * cin * cout)
            # the family name carries the launch size (threads) so that the PMC rows of tools/pmc_traffic.sh match it
            grid = ((tiles + 31) // 32 + 7) // 8 * 8 * (cout // 64) * 512     # csrc/wino_conv2d.hip: 32 tiles x 64 columns per workgroup, tile blocks padded to 8
            return t._timed("conv2d_wino[%d->%d @%d]" % (cin, cout, grid), flops, nbytes, sv["conv2d_wino"], x, u, cout, scale,
                            shift, relu, out, stats)

        def conv2d_wino_wgrad(x, dy, like):
            n, cin, h, w = x.shape
            cout = dy.shape[1]
            tiles = n * ((h + 1) // 2) * ((w + 1) // 2)
            flops = 2.0 * 16 * tiles * cin * cout
            nbytes = 4.0 * (n * h * w * (cin + cout) + 9 * cin * cout)
            ns = max(8, (64 // ((cin // 128) * (cout // 128))) & ~7)          # csrc/wino_wgrad.hip: wgrad_splits
            grid = 4 * ns * (cin // 128) * (cout // 128) * 512
            return t._timed("conv2d_wino_wgrad[%dx%d @%d]" % (cin, cout, grid), flops, nbytes, sv["conv2d_wino_wgrad"], x, dy,
                            like)

        def wino_weight(weight, flip=False, out=None):
            return t._timed("wino_weight", 0.0, 4.0 * weight.numel() * (1 + 16.0 / 9), sv["wino_weight"], weight, flip, out)

        for n, f in dict(conv2d_wino=conv2d_wino, conv2d_wino_wgrad=conv2d_wino_wgrad, wino_weight=wino_weight).items():
            setattr(ops, n, f)
        for n, f in dict(conv_gemm=conv_gemm, conv_gemm_balanced=conv_gemm_balanced, conv_gemm_ring=conv_gemm_ring,
                         conv_ring_plan=conv_ring_plan, conv_plan=conv_plan,
                         conv_group=conv_group, conv_wgrad=conv_wgrad, subm_rulebook=subm_rulebook,
                         conv_rulebook=conv_rulebook, voxelize=voxelize, densify=densify, densify_bwd=densify_bwd,
                         pack_weight=pack_weight).items():
            setattr(ops, n, f)

    def uninstall(self):
        from spx import ops
        for n, f in self._saved.items():
            setattr(ops, n, f)

    def summary(self, nsteps):
        torch.cuda.synchronize()
        fam = {}
        for name, fl, by, e0, e1, _out in self.rec:
            fl = fl() if callable(fl) else fl
            by = by() if callable(by) else by
            d = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, roof_hbm=0.0, roof_mfma=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
            d["roof_hbm"] += by / HBM_PEAK
            d["roof_mfma"] += fl / MFMA_F32_PEAK
        for d in fam.values():
            d["ms_per_step"] = d["ms"] / nsteps
            d["launches_per_step"] = d["launches"] / nsteps
            # fraction of the family's roofline: t_roof = sum over launches of max(bytes / HBM peak, flops / MFMA peak)
            mf = d["roof_mfma"] >= d["roof_hbm"]
            d["bound"] = "mfma" if mf else "hbm"
            d["frac"] = (d["roof_mfma"] if mf else d["roof_hbm"]) / max(d["ms"] * 1e-3, 1e-12)
        return fam


def pmc_traffic(prefix):
    """HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md §HBM) of the kernels whose name starts with
    `prefix`, from the newest profiles/r*_pmc_traffic.json (written by tools/pmc_traffic.sh on the same bench command in
    separate --pmc passes).  None when no such file travels with the repo."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    n = sum(v["launches"] for k, v in d.items() if k.startswith(prefix))
    if n == 0:
        return None
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in d.items() if k.startswith(prefix)) / n


def roofline_of(fam):
    """Dominant hand-written kernel family by time -> the roofline object of the contract."""
    name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
    t = d["ms"] * 1e-3
    if d["roof_mfma"] >= d["roof_hbm"]:
        ach, peak, unit, bound = d["flops"] / t / 1e12, MFMA_F32_PEAK / 1e12, "TFLOP/s", "mfma"
    else:
        ach, peak, unit, bound = d["bytes"] / t / 1e9, HBM_PEAK / 1e9, "GB/s", "hbm"
    rocprof_name, extra_name = "?", None
    if name.startswith("conv_gemm[mfma ") and name.endswith(" ring]"):
        cs, cd = name[len("conv_gemm[mfma "):-len(" ring]")].split("x")
        rocprof_name = "k_conv_ring<%s, %s>" % (cs, cd)
    elif name.startswith("conv_gemm[mfma ") and name.endswith(" balanced]"):
        cs, cd = name[len("conv_gemm[mfma "):-len(" balanced]")].split("x")
        rocprof_name = "k_conv_mfma_pbl%s<%s, %s>" % ("2" if cs == "128" else "", cs, cd)   # one kernel since round 2
    elif name.startswith("conv_gemm[mfma "):
        cs, cd = name[len("conv_gemm[mfma "):-1].split("x")
        rocprof_name = "k_conv_mfma<%s, %s," % (cs, cd)
    elif name.startswith("conv_wgrad[mfma "):
        def _t(c):   # template argument of k_wgrad_mfma for a channel count (csrc/conv_wgrad.hip: spx_conv_wgrad)
            m = (int(c) + 15) // 16
            return 4 if m == 3 else (8 if m > 4 else m)
        ci, co = name[len("conv_wgrad[mfma "):-1].split("x")
        rocprof_name = "k_wgrad_mfma<%d, %d," % (_t(ci), _t(co))       # one API call = + k_wgrad_count + k_wgrad_reduce
    extra = {}
    if name.startswith("conv2d_wino["):
        rocprof_name = "k_wino_conv<1>@" + name.split("@")[1].rstrip("]")
        extra = {"flops_counted": "executed (Winograd domain: 16 multiply-adds per 2x2 tile and channel pair)",
                 "direct_form_equivalent_TFLOP_s": round(2.25 * d["flops"] / t / 1e12, 2)}
    elif name.startswith("conv2d_wino_wgrad["):
        rocprof_name, extra_name = "k_wino_wgrad@" + name.split("@")[1].rstrip("]"), None
        extra = {"flops_counted": "executed (Winograd domain)",
                 "direct_form_equivalent_TFLOP_s": round(2.25 * d["flops"] / t / 1e12, 2)}
    traffic = pmc_traffic(rocprof_name)
    if traffic is not None and extra_name is not None:
        traffic += pmc_traffic(extra_name) or 0.0
    return {"kernel": name, "rocprof_kernel": rocprof_name + (" ...>" if rocprof_name.endswith(",") else "") +
                                                (" + " + extra_name if extra_name else ""),
            "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
            "frac": round(ach / peak, 4),
            "traffic": traffic,
            "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2), "launches_per_step": d["launches_per_step"],
            "algorithmic_flops_per_launch": d["flops"] / d["launches"],
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"], **extra}


def cpu_baseline(cfg_id, mode):
    """One frame of the same workload on the host: oracle sparse ops + torch-CPU dense tail, all cores."""
    from oracle.cpu_backend import use_oracle_backend
    # the GPU box gives one job a 16-core share of a 256-thread host (and no more): use the affinity mask, capped at
    # 16, otherwise BLAS / OpenMP oversubscribe and the measurement is meaningless (256 threads: 132 s per frame)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cores)
    except Exception:
        pass
    dev = torch.device("cpu")
    _cfg, ds, model, opt, sched = build(cfg_id, dev, "f32")
    nb, bs = 3, 4
    batches = make_batches(ds, cfg_id, bs, 0, dev, n=nb)
    warm = make_batches(ds, cfg_id, 1, 7, dev, n=1)[0]
    step = Step(model, opt, sched, 10.0, mode, "f32")
    model.train(mode == "train")
    with use_oracle_backend():
        step(warm)                      # thread pools, oracle library, allocator
        t0 = time.time()
        for b in batches:
            step(b)
        dt = time.time() - t0
    frames = nb * bs
    # BASELINE.json configs[0] (the reference's own CPU-runnable case): KITTI crop, ~16k points / 5k active voxels, batch 1,
    # voxelise + MeanVFE + VoxelBackBone8x + densify forward on the host (BASELINE.md section 5 protocol, bounded: 3 + 12)
    crop = None
    try:
        _c1, ds1, m1, _o1, _s1 = build(1, dev, "f32")
        m1.eval()
        b1 = make_batches(ds1, 1, 1, 0, dev, n=1)[0]
        ts = []
        with use_oracle_backend(), torch.no_grad():
            for i in range(15):
                bd = {"points": b1["points"], "batch_size": 1}
                t1 = time.time()
                for m in m1.module_list[:3]:
                    bd = m(bd)
                if i >= 3:
                    ts.append(time.time() - t1)
        ts = np.sort(np.asarray(ts))
        crop = {"workload": "BASELINE configs[0]: KITTI crop, %d pts / %d voxels, batch 1, voxelise + VoxelBackBone8x + densify "
                            "forward" % (b1["points"].shape[0], bd["voxel_coords"].shape[0]),
                "frames_s_median": round(1.0 / float(np.median(ts)), 3), "ms_median": round(float(np.median(ts)) * 1e3, 1),
                "ms_p10": round(float(ts[int(0.1 * len(ts))]) * 1e3, 1), "ms_p90": round(float(ts[int(0.9 * len(ts)) - 1]) * 1e3, 1),
                "iters": len(ts), "warmup": 3}
    except Exception as e:                                     # never take the headline down
        crop = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port", "configs0_backbone_forward": crop,
            "sample": "%d frames (%d steps of batch %d) of cfg %d, full detector %s; oracle numpy per-offset gather-GEMM-scatter "
                      "sparse ops + torch-CPU dense tail; %.1f s of CPU work after a 1-frame warm-up"
                      % (frames, nb, bs, cfg_id, "fwd+bwd+step" if mode == "train" else "fwd", dt)}


# ------------------------------------------------------------------------------------------ extras (N = 1)

def _time_loop(fn, warmup, iters):
    """seconds per call: `iters` calls between two device synchronisations (host clock, like the headline)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def backbone_work(core, bd):
    """Algorithmic bytes / FLOPs of voxelise + MeanVFE, the 8 rule tables, the 12 sparse convolutions and
    the densification of one forward, from the rule tables the forward left in `bd`; t_roof = sum over kernels of
    max(bytes / HBM peak, FLOPs / fp32-MFMA peak)."""
    import spx
    book = bd["encoded_spconv_tensor"].indice_dict
    npts, c = bd["points"].shape[0], bd["voxel_features"].shape[1]
    nvox = bd["voxel_features"].shape[0]
    items = [("voxelize+meanvfe", 4.0 * npts * c + 4.0 * nvox * (c + 4), 0.0)]
    seen = set()
    for _name, m in core.backbone_3d.named_modules():
        if not isinstance(m, spx.conv.SparseConvolution):
            continue
        rb = book[m.indice_key]
        K = rb.kvol
        if m.indice_key not in seen:
            seen.add(m.indice_key)
            items.append(("rulebook " + m.indice_key, rb.n_in * 16.0 + K * rb.n_out * 4.0 + (0.0 if rb.subm else rb.n_out * 16.0), 0.0))
        P = int((rb.pair[:, :rb.n_out] >= 0).sum().item())
        items.append(("conv %d->%d %s" % (m.in_channels, m.out_channels, m.indice_key),
                      4.0 * (rb.n_in * m.in_channels + rb.n_out * m.out_channels + K * m.in_channels * m.out_channels + K * rb.n_out),
                      2.0 * P * m.in_channels * m.out_channels))
    enc = bd["encoded_spconv_tensor"]
    cells = bd["batch_size"] * int(np.prod(enc.spatial_shape))
    cd = enc.features.shape[1]
    items.append(("densify", 4.0 * (enc.features.shape[0] * cd + cd * cells), 0.0))
    t_roof = sum(max(b / HBM_PEAK, f / MFMA_F32_PEAK) for _n, b, f in items)
    t_hbm = sum(b / HBM_PEAK for _n, b, f in items)
    t_mfma = sum(f / MFMA_F32_PEAK for _n, b, f in items)
    return dict(bytes=sum(b for _n, b, _f in items), flops=sum(f for _n, _b, f in items), t_roof=t_roof,
                bound="mfma" if t_mfma > t_hbm else "hbm")


def rulebook_rates(points, geom, batch, max_voxels, iters=30):
    """Mvoxels/s (input voxels per second) of the 4 submanifold and the 4 strided rule tables of
    VoxelBackBone8x on one batch, each build timed with HIP events at static capacity (no host read in the timed loop)."""
    from pcdet_amd.datasets import synthetic
    from spx import ops
    vox = ops.voxelize(points, geom["point_cloud_range"], geom["voxel_size"], 5, max_voxels, batch_size=batch, batch_col=0,
                       xyz_col=1, feat_col=1, want_voxels=False)
    gs = synthetic.grid_size_of(geom)
    idx, shape = vox["coords"], [int(gs[2]) + 1, int(gs[1]), int(gs[0])]
    strided = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
               ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
    out = {"subm_hash": [], "strided": [], "level": []}

    def ev_time(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / iters

    for j, (k, s, p) in enumerate(strided):
        n = idx.shape[0]
        if j == 0:     # level 1 rows arrive in voxeliser order: the one hash-built table of a forward pass
            out["subm_hash"].append((n, ev_time(lambda: ops.subm_rulebook(idx, batch, shape, (3, 3, 3), unique=True))))   # as the modules call it
        rb = ops.conv_rulebook(idx, batch, shape, k, s, p)
        out["strided"].append((n, ev_time(lambda: ops.conv_rulebook(idx, batch, shape, k, s, p, sync=False))))
        if j < 3:      # strided tables + the submanifold table of the output level from the same rank bitmap, one call
            t = ev_time(lambda: ops.conv_rulebook(idx, batch, shape, k, s, p, sync=False, subm_ksize=(3, 3, 3)))
            out["level"].append((n + rb.n_out, t))
        idx, shape = rb.out_indices, rb.out_shape
    res = {"definition": "input voxels of the table(s) / build time (SURVEY.md 8d); level = strided tables of a stage + the "
                         "submanifold table of its output level built in one call (n_in + n_out voxels)"}
    for kind, rows in out.items():
        res[kind] = round(sum(n for n, _t in rows) / sum(t for _n, t in rows) / 1e6, 1)
        res[kind + "_tables"] = [{"voxels": n, "us": round(t * 1e6, 1), "mvoxels_s": round(n / t / 1e6, 1)} for n, t in rows]
    return res


def extras(core, batches, batch, device, args):
    """Forward-only, backbone-only, rule-table and Waymo figures (N = 1), each timed in this process after the headline."""
    from pcdet_amd.datasets import synthetic
    from pcdet_amd.models.inference import GraphedDetector
    out = {}

    def guard(name, fn):
        try:
            out[name] = fn()
        except Exception as e:                       # an extra must never take the headline down with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    core.eval()
    pts = batches[0]["points"]

    def fwd_modules(mods, points, bsz):
        bd = {"points": points, "batch_size": bsz}
        with torch.no_grad():
            for m in mods:
                bd = m(bd)
        return bd

    def forward():
        t_e = _time_loop(lambda: fwd_modules(core.module_list, pts, batch), 5, 30)
        runner = GraphedDetector(core, batch, int(pts.shape[0] * 1.05) + 64)
        t_g = _time_loop(lambda: runner(pts), 5, 50)
        return {"workload": "cfg 2 full detector forward (eval, f32), batch %d" % batch,
                "eager_frames_s": round(batch / t_e, 1), "eager_ms": round(t_e * 1e3, 3),
                "hipgraph_frames_s": round(batch / t_g, 1), "hipgraph_ms": round(t_g * 1e3, 3),
                "overflow": {k: list(v) for k, v in runner.overflowed().items()}}

    def backbone_forward():
        mods = core.module_list[:3]                  # MeanVFE (+ voxeliser), VoxelBackBone8x, HeightCompression
        bd = fwd_modules(mods, pts, batch)
        work = backbone_work(core, bd)
        t_e = _time_loop(lambda: fwd_modules(mods, pts, batch), 5, 50)
        runner = GraphedDetector(core, batch, int(pts.shape[0] * 1.05) + 64, n_modules=3)
        t_g = _time_loop(lambda: runner(pts), 5, 100)
        best = min(t_e, t_g)
        return {"workload": "cfg 2 voxelise + MeanVFE + VoxelBackBone8x + densify (eval, BN folded, f32), batch %d" % batch,
                "eager_frames_s": round(batch / t_e, 1), "eager_ms": round(t_e * 1e3, 3),
                "hipgraph_frames_s": round(batch / t_g, 1), "hipgraph_ms": round(t_g * 1e3, 3),
                "roofline": {"bound": work["bound"], "t_roof_us": round(work["t_roof"] * 1e6, 2),
                             "t_measured_us": round(best * 1e6, 2), "frac": round(work["t_roof"] / best, 4),
                             "algorithmic_bytes": work["bytes"], "algorithmic_flops": work["flops"],
                             "achieved_GB_s": round(work["bytes"] / best / 1e9, 1),
                             "achieved_TFLOP_s": round(work["flops"] / best / 1e12, 2)}}

    def rulebooks():
        spec = synthetic.CONFIGS[2]
        r = {"cfg2": rulebook_rates(pts, spec["geom"], batch, spec["geom"]["max_voxels"]["train"])}
        b5 = synthetic.make_batch(5, 1)
        p5 = torch.from_numpy(b5["points"]).to(device)
        r["cfg5"] = rulebook_rates(p5, synthetic.CONFIGS[5]["geom"], 1, synthetic.CONFIGS[5]["max_voxels"])
        return r

    def waymo(cfg_id, mode, steps, warmup):
        bsz = synthetic.CONFIGS[cfg_id]["batch"]
        cfg, ds, model, opt, sched = build(cfg_id, device, "f32")
        model.train(mode == "train")
        bts = make_batches(ds, cfg_id, bsz, 0, device, n=2 if mode == "train" else 1)
        caps = None
        if mode == "train" and not args.dynamic:
            from pcdet_amd.models.inference import static_caps_for
            caps = static_caps_for(model, bsz, max(int(b["points"].shape[0]) for b in bts), training=True)
        st = Step(model, opt, sched, cfg.OPTIMIZATION.GRAD_NORM_CLIP, mode, "f32", static_caps=caps)
        i = [0]

        def one():
            st(bts[i[0] % len(bts)])
            i[0] += 1
        t = _time_loop(one, warmup, steps)
        from spx import ops as _ops
        _ops.check_status(device)
        spec = synthetic.CONFIGS[cfg_id]
        return {"workload": "BASELINE configs[%d]: Waymo-shaped, %d pts / %d voxels per frame, batch %d, full SECOND detector %s"
                            % (cfg_id - 1, spec["n_points"], spec["n_active"], bsz,
                               "fwd+bwd+AdamW step" if mode == "train" else "forward"),
                "frames_s": round(bsz / t, 2), "ms_per_step": round(t * 1e3, 3), "steps": steps, "warmup": warmup}

    guard("forward", forward)
    guard("backbone_forward", backbone_forward)
    guard("rulebook_mvoxels_s", rulebooks)
    torch.cuda.empty_cache()
    guard("cfg3_fwd_bwd", lambda: waymo(3, "train", 10, 3))
    torch.cuda.empty_cache()
    guard("cfg5_forward", lambda: waymo(5, "fwd", 20, 5))
    core.train()
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children (one process per GPU, RCCL rendezvous on
    127.0.0.1) from a parent that never initialises the GPU — torch.cuda.device_count() does not — and hand their exit
    code on.  Reference launch: tools/scripts/dist_train.sh:17 (torch.distributed.launch --nproc_per_node=N)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, have))
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the sparse hot path has no CPU fallback)"
    miopen_db = None
    if not args.no_miopen_db and not args.miopen_find:
        from pcdet_amd.utils.miopen_db import use_tuned_db
        miopen_db = use_tuned_db()          # before the first convolution of the process
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_ddp:
        if world == 1:                       # --force-ddp: the N > 1 code path (RCCL process group, DDP) with one rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
        world = dist.get_world_size()        # what RCCL actually joined
        if world != args.gpus:
            sys.stderr.write("bench.py: %d ranks joined, --gpus %d\n" % (world, args.gpus))
            sys.exit(2)
    from pcdet_amd.datasets import synthetic
    global DENSE_LAYOUT
    DENSE_LAYOUT = args.dense_layout
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    batch = args.batch or synthetic.CONFIGS[args.cfg]["batch"]
    cfg, ds, model, optimizer, sched = build(args.cfg, device, args.dense_dtype)
    model.train(args.mode == "train")
    if (world > 1 or args.force_ddp) and args.mode == "train":
        # broadcast_buffers stays at DDP's default (True), as in the reference (tools/train.py:154-155): BatchNorm running
        # statistics are rank 0's on every rank.  wrap_ddp = DistributedDataParallel with the convolution weights' gradients
        # reduced after the end-of-pass join of the weight-gradient stream instead of bucket by bucket (plain DDP serialises
        # that stream with the backward pass: -7 % per GPU, pcdet_amd/utils/ddp_utils.py)
        from pcdet_amd.utils.ddp_utils import wrap_ddp
        model = wrap_ddp(model, late_reduce=not args.plain_ddp, device_ids=[local_rank], bucket_cap_mb=8,
                         gradient_as_bucket_view=True)
    batches = make_batches(ds, args.cfg, batch, rank, device)
    caps = None
    if args.mode == "train" and not args.dynamic:
        from pcdet_amd.models.inference import static_caps_for
        core0 = model.module if hasattr(model, "module") else model
        caps = static_caps_for(core0, batch, max(int(b["points"].shape[0]) for b in batches), training=True)
    step = Step(model, optimizer, sched, cfg.OPTIMIZATION.GRAD_NORM_CLIP, args.mode, args.dense_dtype, static_caps=caps)
    graphed = False
    if caps is not None and world == 1 and args.graph and args.dense_dtype == "f32":
        from pcdet_amd.models.inference import GraphedTrainStep
        step.graphed = GraphedTrainStep(model, batch, max(int(b["points"].shape[0]) for b in batches) + 64,
                                        max(int(b["gt_boxes"].shape[1]) for b in batches),
                                        example=(batches[0]["points"], batches[0]["gt_boxes"]))
        graphed = True

    for i in range(args.warmup):
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    if caps is not None:
        # a level capacity too small for this data would drop rows: the kernels say so in the device status word.  Then the
        # run falls back to exact-size tensors.
        from spx import _lib as _spxlib, ops as _ops0
        for b in batches[args.warmup % len(batches):] + batches[:args.warmup % len(batches)]:
            if args.warmup < len(batches):
                step(b)                                   # make sure every batch of the timed loop has been seen once
        # Ranks see DIFFERENT frames (make_batches seeds by rank), so one rank alone may overflow: the decision is taken by all
        # of them together — MIN over ranks of the status word (error codes are negative) — or the rank that fell back would
        # issue DDP steps (gradient all-reduces) the others never join.
        word = _ops0.status_word(device).clone()
        if world > 1:
            dist.all_reduce(word, op=dist.ReduceOp.MIN)
        rc = int(word.item())
        _ops0.status_word(device).zero_()
        if rc != 0:
            sys.stderr.write("bench.py: device status %d (%s) on some rank -> every rank falls back to exact-size tensors\n"
                             % (rc, _spxlib.load().spx_strerror(rc).decode()))
            caps = None
            step.static_caps = None
            step.graphed = None
            graphed = False
            for i in range(max(args.warmup, 1)):
                step(batches[i % len(batches)])
            torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    from spx import ops as _ops
    _ops.check_status(device)      # a static capacity that overflowed (rows dropped) would invalidate the number: raises

    line = None
    if rank == 0:
        geom = synthetic.CONFIGS[args.cfg]
        line = {
            "metric": "point-cloud frames/sec (%s)" % ("fwd+bwd" if args.mode == "train" else args.mode),
            "value": round(world * batch * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dense_dtype == "f32" else "f32 sparse / %s dense tail (f32 accumulate, f32 head)"
                     % args.dense_dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %s-shaped synthetic, %d pts / %d active voxels per frame, "
                                   "full SECOND detector (VoxelBackBone8x) %s" %
                                   (args.cfg - 1, geom["geom"]["name"].upper(), geom["n_points"], geom["n_active"],
                                    "fwd+bwd+AdamW step" if args.mode == "train" else "forward"),
                       "batch_per_gpu": batch, "global_batch": batch * world, "parallelism": "dp%d" % world,
                       "voxelize_on_gpu": True,
                       "miopen": ("immediate mode + the tuned user find/perf db shipped in miopen_db/ (written by MIOpen's own "
                                  "tuner on these convolutions)" if miopen_db else
                                  ("find mode" if args.miopen_find else "immediate mode, system db")),
                       "execution": ("static row capacities, no host read in the step, rule tables on a second HIP stream, weight "
                                     "gradients on a third"
                                     + ("; forward + backward replayed as one hipGraph, clip + AdamW eager" if graphed else ""))
                                    if caps is not None else "exact-size sparse tensors"},
        }
    # ---- roofline: instrumented extra steps of the same workload (rank 0 prints; all ranks run them so DDP stays in step)
    if not args.no_roofline:
        kt = KernelTimer()
        kt.install()
        # The instrumented steps run the SAME execution mode as the timed ones (static capacities: no host read inside the step,
        # so an event pair holds kernel time only — round 2 ran them on the exact-size path and timed host waits into the
        # index families); the algorithmic FLOPs / bytes are computed from the device-side live counts after each launch has
        # been timed.  Only the hipGraph replay is switched off (it has no per-kernel events) ...
        step.graphed = None
        # ... and every stream is folded into the launch stream: the rule tables normally run on a second stream
        import spx.prebuild as _pb
        _side_saved, _pb.side_stream = _pb.side_stream, (lambda dev: torch.cuda.current_stream(dev))
        # ... and every kernel in stream order: in the timed steps the weight-gradient kernels run on a second stream beside
        # whatever the main stream is doing (spx/functional.py: _off_critical_path), which stretches the kernels they share
        # the chip with; an event pair around a launch would then time the overlap, not the kernel
        import spx.functional as _fn
        _async, _fn._ASYNC_WGRAD = _fn._ASYNC_WGRAD, False
        n_inst = 3
        for i in range(n_inst):
            step(batches[i % len(batches)])
        fam = kt.summary(n_inst)
        kt.uninstall()
        _fn._ASYNC_WGRAD = _async
        _pb.side_stream = _side_saved
        if rank == 0:
            line["roofline"] = roofline_of(fam)
            line["kernels"] = {k: {"ms_per_step": round(v["ms_per_step"], 4), "launches_per_step": v["launches_per_step"],
                                   "GFLOP_per_step": round(v["flops"] / n_inst / 1e9, 3),
                                   "MB_per_step": round(v["bytes"] / n_inst / 1e6, 3),
                                   "bound": v["bound"], "frac": round(v["frac"], 4)} for k, v in fam.items()}
            if args.breakdown:
                for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
                    sys.stderr.write("%-22s %7.3f ms/step  %5.1f launches  %8.2f GFLOP  %8.2f MB  -> %6.2f TFLOP/s %7.1f GB/s\n"
                                     % (k, v["ms_per_step"], v["launches_per_step"], v["flops"] / n_inst / 1e9,
                                        v["bytes"] / n_inst / 1e6, v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                        v["bytes"] / (v["ms"] * 1e-3) / 1e9))
    if rank == 0 and world == 1 and not args.no_extras and args.mode == "train" and args.cfg == 2:
        core = model.module if hasattr(model, "module") else model
        line["extras"] = extras(core, batches, batch, device, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.cfg, args.mode)
    if rank == 0:
        print(json.dumps(line))
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
