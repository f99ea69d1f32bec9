#!/usr/bin/env python
"""Registers, scratch, LDS and occupancy of every gfx950 kernel of libwd_hip.so, as the compiler reports them
(`hipcc -Rpass-analysis=kernel-resource-usage`; needs no GPU).  Writes a markdown table; kernels with scratch or VGPR
spills are listed at the top -- a spill in a hot kernel is a finding.
    python scripts/kernel_resources.py > profiles/<round>_kernel_resources.md"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "wide_deep_amd", "csrc")
FILES = ["hash", "embag", "sparse_update", "sparse_fused", "small_tables", "onehot_path", "dist_exchange", "mlp", "mlp_half",
         "mlp_chain", "mlp_chain8"]


def main():
    tmp = tempfile.mkdtemp()
    procs = []
    for f in FILES:
        err = open(os.path.join(tmp, f + ".txt"), "w")
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                                       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(SRC, f + ".hip"),
                                       "-o", os.path.join(tmp, f + ".o")], stderr=err))
    for p in procs:
        if p.wait() != 0:
            sys.exit("hipcc failed")
    rows = []
    for p in sorted(glob.glob(os.path.join(tmp, "*.txt"))):
        for b in re.split(r"remark: Function Name: ", open(p).read())[1:]:
            g = lambda k: (re.search(k + r": (\S+)", b) or [None, "?"])[1]
            rows.append([os.path.basename(p)[:-4], b.split(" ")[0], g("TotalSGPRs"), g("VGPRs"), g("AGPRs"),
                         g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs Spill"), g("VGPRs Spill"),
                         g(r"LDS Size \[bytes/block\]")])
    names = subprocess.run(["c++filt"] + [r[1] for r in rows], capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        n = n.replace("(anonymous namespace)::", "")
        r[1] = re.sub(r"\(.*\)$", "", n)[:90]
    bad = [r for r in rows if r[5] not in ("0", "?") or r[8] not in ("0", "?")]
    print("# Kernel resources (gfx950, `hipcc -O3 -Rpass-analysis=kernel-resource-usage`): %d kernels\n" % len(rows))
    print("Kernels with scratch memory or VGPR spills: %s\n" % (", ".join("`%s` (%s B/lane scratch, %s VGPR spills)" % (r[1], r[5], r[8]) for r in bad) or "none"))
    print("SGPR spills go to VGPR lanes (no memory traffic).  Dynamic LDS (tower kernel: up to 150 KB, GEMM slabs) is not in the static column.\n")
    print("| file | kernel | SGPR | VGPR | AGPR | scratch B/lane | waves/SIMD | SGPR spills | VGPR spills | static LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | `%s` | %s |" % (r[0], r[1], " | ".join(r[2:])))


if __name__ == "__main__":
    main()
