"""Dev tool: idle gaps of the busiest HIP queue in a rocprofv3 --kernel-trace CSV (which kernels sit either side of every
gap above a threshold), per step of bench.py.  usage: python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    rows = list(csv.DictReader(open(path)))
    byq = defaultdict(list)
    for r in rows:
        byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for q in byq:
        byq[q].sort()
    busy = {q: sum(e - s for s, e, _ in v) for q, v in byq.items()}
    main_q = max(busy, key=busy.get)
    print("queues:", {q: "%d kernels, %.1f ms busy" % (len(v), busy[q] / 1e6) for q, v in byq.items()})
    v = byq[main_q]
    # one period of the step: between the last two launches of a kernel that runs once per step (the fused loss)
    marks = [i for i, k in enumerate(v) if "k_loss_final" in k[2]]
    if len(marks) < 2:
        print("no step marks found")
        return
    step = v[marks[-2]:marks[-1]]
    t0, t1 = step[0][0], step[-1][1]
    busy_main = sum(e - s for s, e, _ in step)
    print("last step on queue %s: %.2f ms wall, %.2f ms busy, %d kernels" % (main_q, (t1 - t0) / 1e6, busy_main / 1e6, len(step)))
    others = [(s, e, n, q) for q, vv in byq.items() if q != main_q for s, e, n in vv if e > t0 and s < t1]
    gaps = []
    for i in range(1, len(step)):
        g = step[i][0] - step[i - 1][1]
        if g > thr * 1e3:
            ov = [n[:40] for s, e, n, q in others if s < step[i][0] and e > step[i - 1][1]]
            gaps.append((g / 1e3, (step[i - 1][1] - t0) / 1e6, step[i - 1][2][:60], step[i][2][:60], ov[:3]))
    print("gaps > %.0f us: %d, total %.2f ms" % (thr, len(gaps), sum(g[0] for g in gaps) / 1e3))
    for g in sorted(gaps, key=lambda x: -x[0])[:40]:
        print("  %7.1f us at %6.2f ms  after %-60s before %-60s | other queues: %s" % g)


if __name__ == "__main__":
    main()
