#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel-trace + stats of the bench command for cfg 2 / 3 / 5; writes the stats CSVs under
# gpurun_out/$1_{cfg2,cfg3,cfg5}.  The program sits directly after `--` (no env / bash hop: the profiler has already
# initialised the GPU).  usage: tools/prof_step.sh <tag> [cfgs...]
tag=${1:-prof}
shift
cfgs=${@:-2 3 5}
cd /tmp && export TMPDIR=/tmp
for c in $cfgs; do
  out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_cfg$c
  mode=train; [ "$c" = "5" ] && mode=fwd
  steps=25; [ "$c" != "2" ] && steps=12
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --cfg $c --mode $mode --steps $steps --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $out.log 2>&1 || exit 1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_cfg${c}_kernel_stats.csv
  rm -rf $out
done
