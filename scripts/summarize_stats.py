#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short, readable table.

usage: summarize_stats.py <kernel_stats.csv> <steps-in-trace> > profiles/<name>.md
Kernel names are shortened (template / lambda noise removed); durations are kept exactly as rocprofv3 wrote them.
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"rocprim::\w+::detail::(\w+?)(_impl|_kernel)?<", name)
    if "rocprim" in name:
        for key in ("onesweep", "histogram", "block_merge", "block_sort", "merge_sort", "scan", "partition"):
            if key in name:
                tag = key
                if key == "block_merge":
                    tag += "(partition)" if "mergepath_partition_config" in name else "(merge)"
                return "rocprim:" + tag
        return "rocprim:" + (m.group(1) if m else "?")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"at::native::", "torch:", name)
    return name[:90]


def main():
    path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            c, t = int(r["Calls"]), int(r["TotalDurationNs"])
            a = rows.setdefault(k, [0, 0, 1 << 62, 0])
            a[0] += c
            a[1] += t
            a[2] = min(a[2], int(r["MinNs"]))
            a[3] = max(a[3], int(r["MaxNs"]))
    tot = sum(v[1] for v in rows.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, (c, t, mn, mx) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f |" % (k, c, t / 1e3, t / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    print("\ntotal kernel time %.1f us" % (tot / 1e3) + (" ; %d profiled steps" % steps if steps else ""))


if __name__ == "__main__":
    main()
