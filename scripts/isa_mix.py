#!/usr/bin/env python
"""Static instruction mix of the hot gfx950 kernels (the compiler's assembly output, `hipcc --cuda-device-only -S`; needs no GPU): MFMA, VALU,
SALU, global/buffer loads and stores, LDS ops, waitcnt / barrier counts and code size per kernel.
    python scripts/isa_mix.py > profiles/<round>_isa_mix.md"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "wide_deep_amd", "csrc")
HOT = ["k_tower_chain", "k_gemm_tn_group", "k_row_update", "k_prefetch_onehot", "k_bucket_onehot", "k_bucket_sort_small", "k_bucket_sort_big", "k_route_unique", "k_bucket_update", "k_bucket_hist", "k_bucket_scatter", "k_bucket_colscan",
       "k_embag_fwd_range", "k_input_layer", "k_hash_bucket", "k_fold_affine_all", "k_mlp_finalize_all", "k_hgemm", "k_gemm<"]
FILES = ["hash", "embag", "sparse_fused", "mlp", "mlp_half", "mlp_chain"]


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic")):
        return "vmem_st"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    tmp = tempfile.mkdtemp()
    rows = []
    for f in FILES:
        asm = os.path.join(tmp, f + ".s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-Wno-unused-function",
                        os.path.join(SRC, f + ".hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL, timeout=900)
        cur, mix = None, None
        for ln in open(asm):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                cur, mix = m.group(1), collections.Counter()
                continue
            if cur and ln.startswith(".Lfunc_end"):          # end of the function body (an early s_endpgm is not the end)
                if mix["other"] or mix["salu"]:
                    rows.append((f, cur, mix))
                cur = None
                continue
            t = ln.strip().split()
            if not cur or not t or t[0].startswith((";", ".", "//")) or t[0].endswith(":"):
                continue
            mix[classify(t[0])] += 1
    if not rows:
        sys.exit("no kernels found")
    names = subprocess.run(["c++filt"] + [r[1] for r in rows], capture_output=True, text=True).stdout.splitlines()
    cols = ["mfma", "valu", "salu", "vmem_ld", "vmem_st", "lds", "smem", "waitcnt", "barrier"]
    print("# Static instruction mix of the hot kernels (gfx950, -O3)\n")
    print("Counts of instructions in the code object, not executed counts; loops are counted once.\n")
    print("| file | kernel | total | " + " | ".join(cols) + " |")
    print("|---|---|---|" + "---|" * len(cols))
    for (f, _, mix), n in zip(rows, names):
        n = re.sub(r"\(.*\)$", "", n.replace("(anonymous namespace)::", "")).replace("void ", "")
        if not any(h in n for h in HOT) or n.endswith(".kd"):
            continue
        print("| %s | `%s` | %d | %s |" % (f, n[:80], sum(mix.values()), " | ".join(str(mix[c]) for c in cols)))


if __name__ == "__main__":
    main()
