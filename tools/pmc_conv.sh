#!/bin/bash
# dev tool: PMC counters for the conv kernels on one layer.
# usage (GPU box): [KERNEL=k_wgrad_mfma WHAT=wgrad LAYER=conv3.1.0 KBARGS="--balanced --group -2"] tools/pmc_conv.sh <tag> [ENV=..]...
tag=$1; shift
KERNEL=${KERNEL:-k_conv_mfma}; WHAT=${WHAT:-fwd}; LAYER=${LAYER:-conv3.1.0}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  env "$@" timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --what $WHAT --layers $LAYER --iters 3 $KBARGS > $out/$name.log 2>&1
  echo "pass $name done"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "$KERNEL" not in k: continue
        agg[k.split("$KERNEL")[1][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "$KERNEL" in k: dur[k.split("$KERNEL")[1][:24]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print("kernel", k, " avg duration under PMC %.1f us" % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c, v in sorted(d.items()):
        print("   %-36s %16.0f per dispatch" % (c, sum(v) / len(v)))
PY
