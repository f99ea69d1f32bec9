#!/usr/bin/env python
"""In-step duration of the input-layer gather (k_prefetch_onehot) from a rocprofv3 --kernel-trace CSV of `python bench.py`: the
launches that run BESIDE a tower launch (their interval overlaps a k_tower_chain8 launch: the steps of the timed graphs) against
the ones that run alone (`roofline_gather_kernel`'s back-to-back launches, priming).  Writes the JSON bench.py reports as
`roofline.rocprof_instep_us` (profiles/r6_gather_instep_rocprof.json) and prints it.
usage: gather_instep_from_trace.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys

ev = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "k_prefetch_onehot" in name or "k_tower_chain8" in name:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "g" if "k_prefetch_onehot" in name else "t"))
ev.sort()
towers = [(s, e) for s, e, k in ev if k == "t"]
ins, solo = [], []
j = 0
for s, e, k in ev:
    if k != "g":
        continue
    while j < len(towers) and towers[j][1] < s:
        j += 1
    beside = any(ts < e and te > s for ts, te in towers[max(j - 1, 0): j + 2])
    (ins if beside else solo).append((e - s) / 1e3)
mean = lambda v: round(sum(v) / len(v), 3) if v else None
out = {"kernel": "k_prefetch_onehot<4, 1>", "what": "rocprofv3 --kernel-trace durations (dispatch start -> completion signal) of the "
       "gather launches of `python bench.py`, split by whether a k_tower_chain8 launch runs at the same time",
       "in_step_launches": len(ins), "in_step_us_mean": mean(ins), "in_step_us_min": round(min(ins), 3) if ins else None,
       "in_step_us_max": round(max(ins), 3) if ins else None, "alone_launches": len(solo), "alone_us_mean": mean(solo)}
print(json.dumps(out))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(json.dumps(out, indent=1) + "\n")
