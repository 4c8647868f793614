#!/bin/bash
# dev tool (GPU box): PMC counters of k_wino_conv from the stand-alone probe (tools/hip/wino_probe.hip built to csrc/build/variants/wino_probe).
# usage: tools/pmc_wino.sh <tag>
tag=$1
bin=$GRAFT_REPO_ROOT/tsm-det-pointcloud-_amd/csrc/build/variants/wino_probe
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_wino_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$name -- $bin > $out/$name.log 2>&1
  echo "pass $name rc $?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_wino_conv" not in k: continue
        agg[r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("grid", k)
    for c, v in sorted(d.items()):
        print("   %-36s %16.0f per dispatch" % (c, sum(v) / len(v)))
PY
