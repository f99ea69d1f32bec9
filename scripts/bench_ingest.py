#!/usr/bin/env python
"""Host TSV ingest alone (no GPU): rows/s of dataset.input_fn on the bundled rows by batch size and parser threads
(WD_INGEST_THREADS), and the single-thread split of one batch job into the two C passes and the Python around them."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wide_deep_amd import dataset as DS

lines = open(os.path.join(ROOT, "tests", "golden", "c1_rows.tsv"), "rb").read().splitlines()
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "t.tsv")
with open(path, "wb") as f:
    for _ in range(400):
        f.write(b"\n".join(lines) + b"\n")
n = 400 * len(lines)
out = {"rows": n, "cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for _ in DS.input_fn(path, None, "eval", 512):
    pass
for bs in (64, 512, 8192):
    for thr in (1, 2, 4, 8, 16):
        os.environ["WD_INGEST_THREADS"] = str(thr)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in DS.input_fn(path, None, "eval", bs):
                pass
            best = min(best, time.perf_counter() - t0)
        out["b%d_t%d" % (bs, thr)] = round(n / best)
os.environ.pop("WD_INGEST_THREADS", None)
ds = DS.CsvDataset(path)
L = DS.ingest_lib()
bestk = {"c": 1e9, "f": 1e9}


class W(object):
    def __init__(self, fn, k):
        self.fn, self.k = fn, k

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        bestk[self.k] = min(bestk[self.k], time.perf_counter() - t)
        return r


class LL(object):
    pass


l2 = LL()
l2.wd_tsv_count, l2.wd_tsv_fill = W(L.wd_tsv_count, "c"), W(L.wd_tsv_fill, "f")
buf, starts, ends = ds._load()
bt = 1e9
for rep in range(200):
    t0 = time.perf_counter()
    ds._batch_c(l2, buf, starts[:512], ends[:512], False)
    bt = min(bt, time.perf_counter() - t0)
out["one_job_512_us"] = {"total": round(1e6 * bt, 1), "count": round(1e6 * bestk["c"], 1), "fill": round(1e6 * bestk["f"], 1)}
print(json.dumps(out))
