"""bench.py end to end on the GPU box: the default command line's instrumented steps wrap every spx.ops entry point, so a new
keyword argument in ops that the wrappers do not pass through breaks the round's headline run (it did once, round 3)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_with_roofline_and_kernels():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                        "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-500:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("point-cloud frames/sec") and d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0 and r["achieved"] > 0 and r["peak"] > 0
    assert r["traffic"] is None or r["traffic"] > 0
    k = d["kernels"]
    for fam in ("subm_rulebook", "conv_rulebook", "conv_gemm[mfma 64x64 ring]", "conv_ring_plan", "densify(+memset)"):
        assert fam in k and k[fam]["ms_per_step"] > 0 and k[fam]["bound"] in ("hbm", "mfma"), fam
