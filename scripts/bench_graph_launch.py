#!/usr/bin/env python
"""What one launch of a multi-step hipGraph costs beyond its steps (run on the GPU box): for graphs of n = 1, 2, 5, 10, 25 steps of
the C2 workload -- T1 = synchronize -> one replay -> synchronize, T2 = the same with two replays back to back.  T2 - T1 = one
graph in steady state (its steps + the gap to the previous graph), 2 T1 - T2 = what the FIRST launch after an idle GPU adds
(host-side launch work before the first kernel starts).  bench.py's timed region at the driver's --steps 20 pays the latter once."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wide_deep_amd import synth, pipeline
from wide_deep_amd.engine import WideDeepEngine
import bench as _bench

B = 8192
spec, mean_len = _bench.make_spec("c2")
eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * (2 * mean_len + 2))
tbs = [synth.TokenBatch(eng.plan, synth.make_raw_batch(eng.plan, B, seed=20260926 + i, mean_len=mean_len)) for i in range(26)]
side = pipeline.warm(eng, tbs)
for n in (1, 2, 5, 10, 25):
    g = pipeline.StepGraph(eng, tbs[:n], stream=side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    res = {}
    for reps in (1, 2, 4):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[reps] = sorted(ts)[len(ts) // 2] * 1e6
    t0 = time.perf_counter()
    for _ in range(20):
        g.graph.replay()
    host = (time.perf_counter() - t0) / 20 * 1e6
    torch.cuda.synchronize()
    steady = (res[4] - res[2]) / 2
    print(json.dumps({"steps_per_graph": n, "T1_us": round(res[1], 1), "T2_us": round(res[2], 1), "T4_us": round(res[4], 1),
                      "steady_graph_us": round(steady, 1), "steady_us_per_step": round(steady / n, 2),
                      "first_launch_extra_us": round(res[1] - steady, 1), "host_us_per_replay_call": round(host, 1)}))

# the driver's timed region: 20 steps as two distinct chained graphs of 10 (bench.py), as one graph of 20, as 4 x 5
def region(graphs, label):
    for _ in range(2):
        for g in graphs:
            g.replay()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"region": label, "us_per_step": round(sorted(ts)[3] * 1e6 / 20, 2), "all": [round(t * 1e6 / 20, 1) for t in ts]}))

def chain(sizes):
    out, ph, j = [], (0, 0), 0
    for n in sizes:
        g = pipeline.StepGraph(eng, [tbs[(j + i) % len(tbs)] for i in range(n)], stream=side, lookahead=tbs[(j + n) % len(tbs)],
                               phase=ph, primed=True)
        out.append(g); ph = g.next_phase; j += n
    return out

region([pipeline.StepGraph(eng, tbs[:10], stream=side), pipeline.StepGraph(eng, tbs[10:20], stream=side)], "2 x 10, independent graphs")
region(chain([10, 10]), "2 x 10, chained")
region(chain([20]), "1 x 20, chained")
region(chain([5, 5, 5, 5]), "4 x 5, chained")
region(chain([2, 6, 12]), "2 + 6 + 12, chained")
