#!/usr/bin/env python
"""Raw kernel sequence of a few consecutive steps from a rocprofv3 --kernel-trace CSV (start / end relative to an anchor launch).
usage: trace_window.py <kernel_trace.csv> [anchor-substring=k_tower_chain] [anchor-index=60] [steps=2]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_tower_chain"
idx = int(sys.argv[3]) if len(sys.argv) > 3 else 60
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ev = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
anchors = [i for i, e in enumerate(ev) if anchor in e[2]]
idx = min(idx, len(anchors) - nsteps - 1)
t0 = ev[anchors[idx]][0]
t1 = ev[anchors[idx + nsteps]][0]
print("%-50s %6s %9s %9s %8s" % ("kernel", "queue", "start us", "end us", "dur us"))
for s, e, n, q, st in ev:
    if t0 - 5000 <= s < t1:
        print("%-50s %6s %9.1f %9.1f %8.1f" % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
