#!/usr/bin/env python
"""What bounds the input-layer gather?    One C2 batch's 212,992 rows gathered out of a 26 M-row table into
a dense [n][16] matrix (the contract's bytes: n * 64 read + n * 4 ids + n * 64 written) by the step's request shape (4 lanes x
16 bytes per row, nontemporal: wd_diag_gather_modes mode 1; mode 6 = its read side alone), varying what the REQUEST STREAM looks like:

  layout   128-byte row records (the step's table: [emb 16 | w z n - | pad], 3.3 GB)  vs  compact 64-byte rows (1.66 GB)
  order    random (as the batch holds them)  vs  ascending row order (what the sorted pair list of wd_bucket_sort would give a
           gather that scatters into x)  vs  ascending + distinct rows of a Zipf(1.05) batch
  batches  1 .. 16 per launch

HIP events over back-to-back launches on fresh id sets; GC_ONLY="<layout>,<order>,<batches>,<mode>" runs one cell (for rocprofv3
--pmc passes: profiles/run_scripts/gpu_r4_gather_pmc.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from wide_deep_amd.capi import call, ptr, load
load()
st = torch.cuda.current_stream().cuda_stream
rows = 26_000_000
g = torch.Generator(device="cuda"); g.manual_seed(7)
base = 212_992
only = os.environ.get("GC_ONLY")
iters_env = int(os.environ.get("GC_ITERS", "0"))


def id_sets(order, n, k):
    out = []
    for i in range(6):
        if order == "zipf_sorted_unique":
            # Zipf(1.05) over 1M ranks per slot, 26 slots (rank -> row by a fixed permutation): the distinct rows of the batch, ascending
            rng = np.random.default_rng(100 + i)
            r = np.arange(1, 1_000_001, dtype=np.float64) ** -1.05
            cdf = np.cumsum(r / r.sum())
            ranks = np.searchsorted(cdf, rng.random(n * k))
            slot = np.repeat(np.arange(26), (n * k + 25) // 26)[: n * k]
            perm_rows = (ranks.astype(np.int64) * 2654435761 % 1_000_000) + slot.astype(np.int64) * 1_000_000
            ids = torch.from_numpy(np.unique(perm_rows).astype(np.int32)).cuda()
        else:
            ids = torch.randint(0, rows, (n * k,), dtype=torch.int32, device="cuda", generator=g)
            if order == "sorted":
                ids = ids.sort().values
        out.append(ids.contiguous())
    return out


print("%-22s %-20s %8s %-10s %9s %10s %12s %10s" % ("layout", "order", "batches", "what", "rows", "us/launch", "us per batch", "frac 8TB/s"))
for rs, lname in ((32, "128 B records"), (16, "64 B compact rows")):
    if only and only.split(",")[0] != str(rs):
        continue
    rec = torch.empty(rows * rs, dtype=torch.float32, device="cuda").normal_()
    for order in ("random", "sorted", "zipf_sorted_unique"):
        if only and only.split(",")[1] != order:
            continue
        for k in (1, 4, 16):
            if only and int(only.split(",")[2]) != k:
                continue
            if order == "zipf_sorted_unique" and k > 1:
                continue
            pools = id_sets(order, base, k)
            n = int(pools[0].numel())
            out = torch.empty(max(p.numel() for p in pools) * 16, dtype=torch.float32, device="cuda")
            for mode in (1, 6):
                if only and int(only.split(",")[3]) != mode:
                    continue
                run = lambda i: call("wd_diag_gather_modes", ptr(rec), rs, ptr(pools[i % 6]), int(pools[i % 6].numel()), mode, ptr(out), st)
                for i in range(4):
                    run(i)
                torch.cuda.synchronize()
                iters = iters_env or (40 if k <= 4 else 12)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(iters):
                    run(i)
                e1.record(); e1.synchronize()
                us = e0.elapsed_time(e1) / iters * 1e3
                alg = n * 64 + n * 4 + (0 if mode == 6 else n * 64)
                print("%-22s %-20s %8d %-10s %9d %10.2f %12.2f %10.3f" % (lname, order, k, "gather" if mode == 1 else "reads only", n, us, us / k,
                                                                        alg / us / 1e3 / 8000), flush=True)
            del pools, out
    del rec
