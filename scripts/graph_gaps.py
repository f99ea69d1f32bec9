#!/usr/bin/env python
"""Cycle times (tower start -> next tower start) from a rocprofv3 kernel trace, and what happens at the long ones (graph
boundaries).  usage: graph_gaps.py <kernel_trace.csv> [last N towers = 60]"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tw = [i for i, e in enumerate(ev) if "k_tower_chain" in e[2]][-N:]
cyc = [(ev[b][0] - ev[a][0]) / 1e3 for a, b in zip(tw, tw[1:])]
med = sorted(cyc)[len(cyc) // 2]
print("tower-to-tower cycle: median %.1f us; all: %s" % (med, " ".join("%.0f" % c for c in cyc)))
for k, c in enumerate(cyc):
    if c > med * 1.12:
        a, b = tw[k], tw[k + 1]
        print("-- long cycle %.1f us (median %.1f): kernels between the two towers" % (c, med))
        t0 = ev[a][0]
        for s, e, n in ev[a:b + 1]:
            print("   %-42s %8.1f %8.1f" % (n, (s - t0) / 1e3, (e - t0) / 1e3))
