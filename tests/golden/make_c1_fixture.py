#!/usr/bin/env python
"""Builds tests/golden/c1_rows.tsv from the reference's bundled click log (run in the build container only).

Rows 0..(N-1) of data/train/train1 plus every positive row of the file (6 of 5000), so that a 512-row batch of real
tokens (multi-valued fields, '-' NAs, 32-hex / UUID / up-to-54-byte tokens) travels to the GPU box, where
/root/reference does not exist.  No expected outputs exist for these rows in the reference: the file is
an INPUT fixture; parity on it is GPU-vs-oracle.
"""
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/train/train1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 560
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_rows.tsv")

with open(SRC, "rb") as f:
    lines = f.readlines()
keep = list(range(N)) + [i for i, ln in enumerate(lines) if i >= N and ln.split(b"\t", 1)[0] == b"1"]
with open(OUT, "wb") as f:
    for i in keep:
        f.write(lines[i])
print("wrote %d rows (%d positives) to %s" % (len(keep), sum(lines[i].startswith(b"1\t") for i in keep), OUT))
