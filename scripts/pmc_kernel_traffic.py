#!/usr/bin/env python
"""HBM-side traffic of named kernels from two rocprofv3 --pmc passes (L2 -> fabric requests by size, the calibration of
profiles/r5_gather_counter_calibration.md):  bytes = 128 RDREQ_128B + 64 RDREQ_64B + 32 RDREQ_32B + 64 WRREQ_64B + 32 (WRREQ - WRREQ_64B).

usage: pmc_kernel_traffic.py <dir of the read pass> <dir of the write pass> <kernel substring> [<kernel substring> ...]
(each dir holds the *counter_collection.csv of `rocprofv3 --kernel-trace --pmc <counters> --output-format csv`).  Prints, per kernel,
the mean per launch over the launches of the second half of the run (steady state) and the line counts."""
import csv
import glob
import sys


def collect(d, names):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    out = {n: {} for n in names}
    for r in csv.DictReader(open(f[0])):
        for n in names:
            if n in r["Kernel_Name"]:
                out[n].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def mean_tail(v):
    v = v[len(v) // 2:]
    return sum(v) / max(len(v), 1)


def main():
    rd_dir, wr_dir, names = sys.argv[1], sys.argv[2], sys.argv[3:]
    rd, wr = collect(rd_dir, names), collect(wr_dir, names)
    for n in names:
        g = {k: mean_tail(v) for k, v in list(rd[n].items()) + list(wr[n].items())}
        if not g:
            print("%-28s no launches found" % n)
            continue
        r128, r64, r32 = g.get("TCC_EA0_RDREQ_128B_sum", 0.0), g.get("TCC_EA0_RDREQ_64B_sum", 0.0), g.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        w, w64 = g.get("TCC_EA0_WRREQ_sum", 0.0), g.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        rb, wb = 128 * r128 + 64 * r64 + 32 * r32, 64 * w64 + 32 * (w - w64)
        print("%-28s launches %4d | reads %.3e x128 %.3e x64 %.3e x32 = %7.1f MB | writes %.3e (64 B: %.3e) = %7.1f MB | total %7.1f MB"
              % (n, len(next(iter(rd[n].values()), [])), r128, r64, r32, rb / 1e6, w, w64, wb / 1e6, (rb + wb) / 1e6))


if __name__ == "__main__":
    main()
