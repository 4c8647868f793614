"""dev tool: idle gaps of the GPU inside the timed steps, from a rocprofv3 --kernel-trace csv of
`bench.py --no-roofline --no-cpu-baseline` (kernels sorted by start; a gap = start - latest end so far)."""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"at::native::", "", n)
    return n.split("(")[0][:70]

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# steps are delimited by the optimizer kernel (multi_tensor_apply ... FusedOptimizer): keep the last 12 steps
opt = [i for i, e in enumerate(ev) if 'FusedOptimizerTensorListMetadata' in e[2]]
per_step = 3
bounds = opt[per_step - 1::per_step]
lo, hi = bounds[-13], bounds[-1]
ev = ev[lo + 1:hi + 1]
steps = 12
span = ev[-1][1] - ev[0][0]
end, tot_gap, busy = ev[0][0], 0, 0
gaps, gapn = collections.Counter(), collections.Counter()
big = []
prev = ev[0][2]
for s, e, n in ev:
    if s > end:
        g = s - end
        tot_gap += g
        key = short(n)
        gaps[key] += g
        gapn[key] += 1
        if g > 30000:
            big.append((g, short(prev), key))
    busy += max(0, e - max(s, end))
    if e > end:
        end = e
    prev = n
print("per step: span %.3f ms, busy %.3f ms, idle %.3f ms" % (span / steps / 1e6, busy / steps / 1e6, tot_gap / steps / 1e6))
for k, v in gaps.most_common(18):
    print("%7.3f ms/step  n/step %5.1f  avg %6.1f us  before %s" % (v / steps / 1e6, gapn[k] / steps, v / gapn[k] / 1e3, k))
print("gaps > 30 us (per 12 steps): %d" % len(big))
agg = collections.Counter()
for g, a, b in big:
    agg[(a, b)] += g
for (a, b), g in agg.most_common(12):
    print("  %7.3f ms/step  after %s -> before %s" % (g / steps / 1e6, a, b))
