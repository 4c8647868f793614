#!/bin/bash
# dev tool: FETCH_SIZE / WRITE_SIZE of the balanced conv kernel on one layer for several library builds / settings.
# usage (GPU box): tools/pmc_fetch_ab.sh "<label>|<ENV=.. ENV=..>" ...
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_ab
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "$@"; do
  label=${spec%%|*}; envs=${spec#*|}
  for c in FETCH_SIZE WRITE_SIZE; do
    env $envs SPX_CONV_BALANCED_SHAPES=64x64 timeout -k 5 120 rocprofv3 --pmc $c --output-format csv -d $out/$i.$c -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --what fwd --balanced --group ${GROUPW:--1} --layers ${LAYER:-conv3.1.0} --iters 3 > $out/$i.$c.log 2>&1
  done
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/$i.*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_conv_mfma_pbl" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
# dispatches alternate: warm-up + timed ungrouped, then grouped ...: report min / max per dispatch of 2F+W
f = {d: sum(v) for d, v in acc["FETCH_SIZE"].items()}
print("$label: pbl dispatches FETCH_SIZE (raw KB):", sorted(set(int(x) for x in f.values()))[:8])
PY
  i=$((i+1))
done
