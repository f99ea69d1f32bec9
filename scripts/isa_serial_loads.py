#!/usr/bin/env python
"""Inner loops whose body is `load -> s_waitcnt vmcnt(0) -> use`: one dependent memory round trip per iteration.
Compiles every csrc/*.hip to gfx950 assembly (hipcc -S, no GPU needed) and lists (file, kernel, loop label, loads, length).
Round 4 found the step's two cheapest wins this way (the dense tail's logits-layer sum: 14 -> 6 us; one-workgroup reductions
of the loss / bias gradient).  A hit is a candidate, not a verdict: a per-occurrence chain inside thousands of resident
wavefronts is hidden by occupancy; a chain in ONE workgroup on the step's critical path is not.
usage: python scripts/isa_serial_loads.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wide_deep_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
tmp = tempfile.mkdtemp(prefix="wd_isa_")
for f in files:
    out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        "-I" + CSRC, f, "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        print(os.path.basename(f), "did not compile:", r.stderr.strip().splitlines()[-1:])
        continue
    lines = open(out).read().splitlines()
    kern = None
    for idx, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", m.group(1))[:64]
        m = re.match(r"^(\.LBB\d+_\d+):\s*;.*Inner Loop Header", l)
        if not m:
            continue
        lab, body, closed = m.group(1), [], False
        for j in range(idx + 1, min(idx + 400, len(lines))):
            body.append(lines[j])
            if re.search(r"s_cbranch_\w+ " + re.escape(lab) + r"\b", lines[j]):
                closed = True
                break
        if not closed:
            continue
        loads = sum(1 for b in body if re.search(r"\b(global|buffer|flat)_load", b))
        waits = sum(1 for b in body if re.search(r"s_waitcnt.*vmcnt\(0\)", b))
        if loads and waits and loads <= 3 and len(body) < 80 and "rocprim" not in kern:
            print("%-18s %-66s %-10s loads %d, %d instructions" % (os.path.basename(f), kern, lab, loads, len(body)))
