#!/bin/bash
# dev tool: PMC counters for the conv kernels on one layer.  usage (on the GPU box): tools/pmc_conv.sh <tag> [env...]
# counters collected in their own runs (no tracing), per MI355X_MICROARCH.md "rocprofv3 PMC slots".
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  env "$@" rocprofv3 --pmc $set --output-format csv -d $out/$name -- timeout -k 5 120 python3 $GRAFT_REPO_ROOT/tools/kbench.py --what fwd --layers conv3.1.0 --iters 3 > $out/$name.log 2>&1; echo "pass $name done"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv_mfma" not in k: continue
        short = k.split("k_conv_mfma")[1][:24]
        agg[short][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVES","FETCH_SIZE","TA_TA_BUSY","TCC_REQ_sum","SQ_INSTS_LDS","WRITE_SIZE"): cnt[(short, r["Counter_Name"])] += 1
for k, d in agg.items():
    print("kernel", k)
    for c, v in sorted(d.items()):
        n = max(1, max([cnt[(k, x)] for x in ("SQ_WAVES","FETCH_SIZE","TA_TA_BUSY","TCC_REQ_sum","SQ_INSTS_LDS","WRITE_SIZE")] or [1]))
        print("   %-36s %16.0f total" % (c, v))
    print("   dispatch counts", {c: n for (kk, c), n in cnt.items() if kk == k})
PY
