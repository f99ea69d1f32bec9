#!/usr/bin/env python
"""Per step of a rocprofv3 --kernel-trace CSV (a step = one anchor launch to the next): its duration and the hardware queue every
named kernel ran on -- which graph branches shared a queue in which step.
usage: step_queues.py <kernel_trace.csv> [anchor-substring=k_tower_chain] [first=20] [count=30] [kernels=comma,separated,substrings]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_tower_chain"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 20
count = int(sys.argv[4]) if len(sys.argv) > 4 else 30
names = (sys.argv[5] if len(sys.argv) > 5 else "k_row_update,k_gemm_tn,k_feat_emit,k_bucket_hist,k_small_bwd,k_fingerprint64").split(",")
ev = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
ev.sort()
anchors = [i for i, e in enumerate(ev) if anchor in e[2]]
print("step  us      %s  " % anchor + "  ".join("%s(q start dur)" % n for n in names))
for k in range(first, min(first + count, len(anchors) - 1)):
    a, b = anchors[k], anchors[k + 1]
    t0 = ev[a][0]
    row = ["%4d %7.1f  q%s" % (k, (ev[b][0] - t0) / 1e3, ev[a][3])]
    for n in names:
        hit = [e for e in ev[a:b] if n in e[2]]
        row.append("q%s %6.1f %6.1f" % (hit[0][3], (hit[0][0] - t0) / 1e3, (hit[0][1] - hit[0][0]) / 1e3) if hit else "-")
    print("   ".join(row))
