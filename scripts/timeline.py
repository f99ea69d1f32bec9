#!/usr/bin/env python
"""Step timeline from a rocprofv3 --kernel-trace CSV: for the steady-state steps (between consecutive k_tower_chain starts)
print every kernel's mean start offset and mean duration relative to the tower launch of its step.

usage: timeline.py <kernel_trace.csv> [anchor-substring=k_tower_chain] [skip-first=40]"""
import csv
import sys
import collections


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_gemm_tn_group"
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    ev.sort()
    anchors = [i for i, e in enumerate(ev) if anchor in e[2]]
    # steady state: anchors whose distance to the next anchor is within 2x the median period
    per = sorted(ev[anchors[i + 1]][0] - ev[anchors[i]][0] for i in range(len(anchors) - 1))
    med = per[len(per) // 2]
    steps = [(anchors[i], anchors[i + 1]) for i in range(skip, len(anchors) - 1)
             if ev[anchors[i + 1]][0] - ev[anchors[i]][0] < 1.6 * med]
    print("median step period %.1f us over %d anchors; %d steady steps used" % (med / 1e3, len(anchors), len(steps)))
    agg = collections.OrderedDict()
    for a, b in steps:
        t0 = ev[a][0]
        seen = collections.Counter()
        # kernels that START in [t0 - 0.6 period, t0 + 0.4 period) belong to this step's neighbourhood
        for s, e, n in ev[max(0, a - 30): b + 30]:
            if t0 - 0.6 * med <= s < t0 + 0.4 * med:
                key = (n.split("(")[0][-60:], seen[n])
                seen[n] += 1
                agg.setdefault(key, []).append((s - t0, e - s))
    rows = sorted(((sum(x[0] for x in v) / len(v), k, v) for k, v in agg.items()))
    print("%-64s %5s %9s %9s %9s" % ("kernel", "n", "start us", "dur us", "end us"))
    for st, k, v in rows:
        if len(v) < 0.5 * len(steps):
            continue
        d = sum(x[1] for x in v) / len(v)
        print("%-64s %5d %9.1f %9.1f %9.1f" % (k[0] + ("#%d" % k[1] if k[1] else ""), len(v), st / 1e3, d / 1e3, (st + d) / 1e3))


if __name__ == "__main__":
    main()
