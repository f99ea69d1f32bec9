#!/usr/bin/env python
"""bench.py -- examples/sec of the Wide&Deep train step on MI355X (BASELINE.json metric).

A "step" = one full train step of the hot path over one synthetic Criteo-shaped batch that is
already resident in HBM as raw string tokens: hash -> embedding-bag gather + wide sum -> BN/ReLU
tower (fp32 MFMA) -> sigmoid-CE (batch SUM) -> backward -> Adagrad (dense + embedding rows) + FTRL
(wide rows).  Nothing is skipped inside the timed region.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (embedding-gather kernel,
HBM-bound; achieved = algorithmic bytes / measured kernel time) and `cpu_baseline` (the CPU oracle
timed on this box's host cores on a bounded sample of the same workload; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # CPU-baseline leg: no spin-waiting across OpenMP runtimes

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8192, help="examples per GPU per step")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c4-nocross", "c5"])
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--pool", type=int, default=32, help="distinct resident batches cycled through")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of hipGraph replay")
    ap.add_argument("--steps-per-graph", type=int, default=32,
                    help="train steps captured into one hipGraph (single GPU; each step on its own resident batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline traffic = null)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of batch 0 before the timed region")
    ap.add_argument("--dump-stamps", default=None, metavar="PATH",
                    help="write the per-launch clock stamps `roofline` is computed from (first workgroup start, last store landed)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="diagnostics: run the row-sharded engine even with one rank (one-rank RCCL group)")
    ap.add_argument("--cpu-steps", type=int, default=50)
    ap.add_argument("--probe-capture", action="store_true",
                    help="(internal) child of a rank: capture and replay a small sharded step graph with the collectives in it")
    ap.add_argument("--cpu-only", action="store_true",
                    help="print only the cpu_baseline object of the workload (what an N > 1 run launches as a child of rank 0)")
    ap.add_argument("--ids-input", action="store_true", help="feed pre-hashed ids (skips the hash kernel)")
    ap.add_argument("--repeats", type=int, default=9,
                    help="the K timed steps are measured this many times back to back (each bracketed by barrier + synchronize); "
                         "`value` is the median, all of them are printed in `repeats_ms_per_step`")
    ap.add_argument("--slack", type=float, default=0.0,
                    help="sharded step: per-peer segment capacity over the expectation (default: 1.1 uniform ids, 1.2 with the "
                         "sender-side unique, 2.5 for skewed ids sent per occurrence); check_overflow reports a segment that was too small")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch examples per GPU; strong: --batch is the GLOBAL batch, split evenly over the ranks")
    ap.add_argument("--tower-dtype", default=None, choices=["fp32", "fp16"],
                    help="GEMM operand type of the tower (default: fp32; c5: fp16 = BASELINE configs[4])")
    return ap.parse_args()


C4_CROSSES = ((0, 1), (2, 3, 4))


def make_spec(cfg):
    from wide_deep_amd.plan import criteo_spec
    if cfg == "c2":
        # BASELINE.json configs[1]: 13 dense + 26 sparse slots, 1M buckets, emb 16, Dnn [256,128,64]
        return criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="simple"), 1
    if cfg == "c3":
        # BASELINE.json configs[2] table: ONE logical 100M-row table (26 slots x 3,846,154 rows), C2 otherwise.  With
        # --gpus N it is row-sharded over the ranks; with one GPU all 100M rows (6.4 GB + 6.4 GB Adagrad + 1.6 GB
        # wide) are resident in its 288 GB
        return criteo_spec(n_dense=13, n_sparse=26, buckets=3_846_154, dim=16, hidden=(256, 128, 64), mode="simple"), 1
    if cfg == "c5":
        # BASELINE.json configs[4]: deep-only DenseDnn [1024,512,256,128], emb_dim 64, fp16 MFMA tower (fp32 embeddings);
        # --tower-dtype fp32 runs the same shape on the exact-fp32 MFMA tower
        return criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=64, hidden=(1024, 512, 256, 128),
                           mode="dense", model_type="deep"), 1
    # BASELINE.json configs[3] on one GPU: multi-hot (avg 5 ids/slot), ResDnn, weight column -- and, "c4", its 200-bucket
    # crossed columns over 2 and 3 slots ("200-bucket crosses over 2-3 slots each"; python/lib/build_estimator.py:
    # 138-155; embedding_dim(200) = 4, so the deep input mixes two embedding widths).  "c4-nocross" = the workload rounds 1-4
    # ran under the name c4
    return criteo_spec(n_dense=13, n_sparse=26, buckets=1_000_000, dim=16, hidden=(256, 128, 64), mode="resnet",
                       use_weight_column=True, crosses=C4_CROSSES if cfg == "c4" else (), cross_buckets=200), 5


def event_time_ms(fn, iters):
    """GPU time of `iters` back-to-back calls on the current stream (HIP events on that stream)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def gather_alg_bytes(plan, bt, dim, nslots, whole_layer=False):
    """SURVEY 8(d) contract: nnz*D*4 (row reads) + nnz*4 (ids) + (B*S+1)*4 (offsets) + B*S*D*4 (pooled write).
    whole_layer (models with several embedding widths, BASELINE configs[3] with its crossed columns): "all" = every embedded
    column with ITS width and ITS ids (a crossed column's 25-125 ids per bag read 16-byte rows, not the 64-byte rows of the hash
    slots); True = the columns of width `dim` only (the embedding-bag launch of that width)."""
    if not whole_layer or len(plan.emb_groups) <= 1:
        return bt.nnz * dim * 4 + bt.nnz * 4 + (bt.B * plan.S + 1) * 4 + bt.B * nslots * dim * 4
    offs = bt.bag_offs[: bt.B * plan.S + 1].to(torch.int64).cpu()
    lens = (offs[1:] - offs[:-1]).reshape(bt.B, plan.S).sum(dim=0)
    groups = plan.emb_groups.items() if whole_layer == "all" else [(dim, plan.emb_groups[dim])]      # one launch = one width
    rows = sum(int(lens[i]) * int(d) * 4 for d, sl in groups for i in sl)
    ids = sum(int(lens[i]) for d, sl in groups for i in sl) * 4
    pooled = sum(bt.B * len(sl) * int(d) * 4 for d, sl in groups)
    return rows + ids + (bt.B * plan.S + 1) * 4 + pooled


def pmc_traffic(args, bt, plan):
    """HBM traffic of the stand-alone gather kernel, measured BY THIS RUN: two child passes of the same kernel on the same
    workload under `rocprofv3 --kernel-trace --pmc` (separate passes, as the microarch guide prescribes), with the L2 -> fabric
    REQUEST counters by size instead of the derived FETCH_SIZE / WRITE_SIZE:
        bytes = 128 RDREQ_128B + 64 RDREQ_64B + 32 RDREQ_32B + 64 WRREQ_64B + 32 (WRREQ - WRREQ_64B).
    Why (profiles/r5_gather_counter_calibration_pass{1,2}.txt, calibration on the gather's own access pattern): FETCH_SIZE is
    RDREQ x 64 B, but on gfx950 a read request is a whole 128-byte line -- a 256 MiB streaming copy issues exactly 2^21 of them, all
    in RDREQ_128B -- and a random row costs ONE such request whether 64 or 128 bytes of its record are needed (1.03 per row either
    way; RDREQ_64B ~ 0).  Rounds 1-4 took FETCH_SIZE at face value for the random rows (x 1.0) and reported 1.04 x the algorithmic
    bytes; the rows really move 128 B each.  The 256 MiB copy of the same pass is the check: it must read back x 1.000 and write
    x 1.000 by this formula.  None on any failure."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    env = dict(os.environ, GATHER_ITERS="20", GATHER_POOL="8", GATHER_CONFIG="c4-nocross" if args.config == "c4" else args.config,
               GATHER_BATCH=str(args.batch), GATHER_DIST=args.dist, TMPDIR="/tmp")
    passes = (("TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum"), ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))
    val, cal = {}, {}
    for ctrs in passes:
        d = tempfile.mkdtemp(prefix="wd_pmc_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + list(ctrs) + ["--output-format", "csv", "-d", d, "-o", "pmc", "--",
                            sys.executable, os.path.join(ROOT, "scripts", "bench_gather.py")], env=env, cwd="/tmp",
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            g, c = {k: [] for k in ctrs}, {k: [] for k in ctrs}
            for r in csv.DictReader(open(f[0])):
                if r["Counter_Name"] not in g:
                    continue
                if "k_embag_fwd" in r["Kernel_Name"] or "k_prefetch_onehot" in r["Kernel_Name"]:
                    g[r["Counter_Name"]].append(float(r["Counter_Value"]))
                elif "copy" in r["Kernel_Name"].lower():
                    c[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k in ctrs:
                val[k] = sum(g[k][-20:]) / len(g[k][-20:])          # requests per launch
                cal[k] = max(c[k]) if c[k] else None                 # requests of the 256 MiB copy
        except Exception as e:
            return None, "PMC pass %s failed: %s" % ("+".join(ctrs), str(e)[:120])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd = 128 * val["TCC_EA0_RDREQ_128B_sum"] + 64 * val["TCC_EA0_RDREQ_64B_sum"] + 32 * val["TCC_EA0_RDREQ_32B_sum"]
    wr = 64 * val["TCC_EA0_WRREQ_64B_sum"] + 32 * (val["TCC_EA0_WRREQ_sum"] - val["TCC_EA0_WRREQ_64B_sum"])
    chk = ""
    if all(v is not None for v in cal.values()):
        crd = 128 * cal["TCC_EA0_RDREQ_128B_sum"] + 64 * cal["TCC_EA0_RDREQ_64B_sum"] + 32 * cal["TCC_EA0_RDREQ_32B_sum"]
        cwr = 64 * cal["TCC_EA0_WRREQ_64B_sum"] + 32 * (cal["TCC_EA0_WRREQ_sum"] - cal["TCC_EA0_WRREQ_64B_sum"])
        chk = "; the 256 MiB copy of the same passes reads back x%.3f / writes x%.3f by the same formula" % (crd / float(1 << 28), cwr / float(1 << 28))
    nrow = max(bt.nnz, 1)
    return int(rd + wr), ("measured by this run: rocprofv3 --pmc passes of the kernel alone, L2 -> fabric requests by size: reads "
                          "%.0f x 128 B + %.0f x 64 B + %.0f x 32 B = %d B (%.2f requests of 128 B per gathered row: a random row costs a "
                          "whole 128-byte line), writes %d B%s" % (val["TCC_EA0_RDREQ_128B_sum"], val["TCC_EA0_RDREQ_64B_sum"],
                                                                  val["TCC_EA0_RDREQ_32B_sum"], int(rd), val["TCC_EA0_RDREQ_128B_sum"] / nrow,
                                                                  int(wr), chk))


def gather_kernel_roofline(eng, batches, args, iters=200):
    """The embedding-gather kernel of the C ABI as its own launch (wd_embag_fwd_range) on the resident batches: algorithmic
    bytes / mean launch duration (HIP events on the launch stream, the pool cycled so ids are never cache-warm)."""
    plan = eng.plan
    tw0 = eng.towers[0]
    ld = tw0["layout"].ld
    xp = tw0["act"].data_ptr() + 4 * tw0["layout"].seg_start[0]
    st = torch.cuda.current_stream().cuda_stream
    (dim, gs), = list(eng.group_slots.items())[:1]

    pf = bool(getattr(eng, "prefetch", False) and batches[0].one_hot)

    def run(i):
        if pf:       # the input layer as the step launches it: records -> x, wide weights, numeric columns
            eng._prefetch_input(batches[i % len(batches)], st, i % eng.n_act)
        else:
            eng.embag_fwd(dim, gs, batches[i % len(batches)], xp, ld, st)

    for i in range(10):
        run(i)
    torch.cuda.synchronize()
    ms = event_time_ms(run, iters)
    bt = batches[0]
    alg = gather_alg_bytes(plan, bt, dim, gs.numel(), whole_layer=True)
    gbs = alg / (ms * 1e-3) / 1e9
    traffic, src = (None, "skipped (--no-pmc)") if args.no_pmc else pmc_traffic(args, bt, plan)
    kname = ("k_prefetch_onehot<%d, 1> on %d-byte row records" % (dim // 4, 4 * eng.rec_stride) if pf
             else "k_embag_fwd<%d> on %d-byte row records" % (dim // 4, 4 * eng.rec_stride) if eng.rec is not None
             else "k_embag_fwd_range<%d, 2, %s>" % (dim // 4, "true" if bt.one_hot else "false"))
    return {"bound": "hbm", "kernel": kname,
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": int(alg),
            "avg_launch_us": round(ms * 1e3, 2),
            "note": ("the input-layer launch of the step (wd_prefetch_onehot: one 128-byte record per id -> embedding row into x, wide "
                     "weight into the per-occurrence list, numeric columns) timed alone, back to back over the resident pool; the "
                     "same launch inside the pipelined step, beside the tower of the previous batch, is `roofline`" if pf else
                     "the embedding-bag gather of the C ABI timed as its own launch on the engine's table layout; multi-hot "
                     "batches take it (or k_input_layer) inside the step")}


def rocprof_instep_record():
    """The tracked rocprofv3 --kernel-trace figure for the same launch in the same graphs (profiles/r6_gather_instep_rocprof.json,
    written by scripts/gather_instep_from_trace.py from the trace of this command on the box of profiles/README.md): bench.py cannot
    trace itself, so the line carries the committed number next to the stamps it measures live."""
    path = os.path.join(ROOT, "profiles", "r6_gather_instep_rocprof.json")
    try:
        return json.load(open(path))
    except Exception:
        return None


def gather_instep_roofline(eng, dev_batches, step_eager, steps=16, runner=None, span=None, dump=None):
    """The gather AS IT RUNS IN THE STEP.  One-id-per-bag batches (C2 / C3): the input layer is the first phase of
    k_tower_chain; every workgroup stores the chip-wide realtime clock (100 MHz) at its start and when its x tile is complete
    (wd_chain_opts_t.tile_stamps), and the phase's duration is  max(tile complete) - min(start)  over all workgroups of a
    launch, averaged over `steps` full train steps on different resident batches.  Other batches: k_input_layer (its own
    launch in the step) timed with HIP events.  achieved = algorithmic gather bytes / that duration."""
    plan = eng.plan
    bt0 = dev_batches[0].batch
    (dim, gs), = list(eng.group_slots.items())[:1]
    alg = gather_alg_bytes(plan, bt0, dim, gs.numel(), whole_layer="all")
    side = torch.cuda.Stream()
    if span is not None and runner is not None and runner.multis:
        # the input layer is its own launch (wd_prefetch_onehot), issued one step ahead beside the tower of the previous batch:
        # every workgroup stores the chip-wide realtime clock (100 MHz) at its start and, once its stores have completed, at its
        # end (per activation buffer).  Measured on THE GRAPHS THAT WERE TIMED (pipeline.StepRunner, captured with the stamp
        # buffer attached): after a replay the buffer holds the stamps of the last launch into each activation buffer
        durs, rows = [], []
        for i in range(steps):
            runner.run(runner.spg)
            torch.cuda.synchronize()
            v = span.cpu()
            for p in range(v.shape[0]):
                t0, t1 = int(v[p, :, 0].min()), int(v[p, :, 1].max())
                durs.append((t1 - t0) / 100.0)
                rows.append((i, p, t0, t1, int(v[p, :, 0].max()) - t0))
        us = sum(durs) / len(durs)
        if dump:
            with open(dump, "w") as f:
                f.write("# k_prefetch_onehot inside the timed graphs: chip-wide 100 MHz realtime clock, one line per launch (the last launch\n"
                        "# into each of the %d activation buffers after a replay of %d steps)\n"
                        "# replay  buffer  first_workgroup_start  last_store_landed  duration_us  start_skew_of_the_workgroups_us\n"
                        % (v.shape[0], runner.spg))
                for i, p, t0, t1, skew in rows:
                    f.write("%4d %4d %16d %16d %8.2f %8.2f\n" % (i, p, t0, t1, (t1 - t0) / 100.0, skew / 100.0))
        kernel = "k_prefetch_onehot<%d, 1> (the input layer of batch t+1, launched beside the tower of batch t)" % (dim // 4)
        how = ("realtime-clock stamps of the launch's workgroups, first start -> last end (stores completed), in the chained "
               "%d-step hipGraphs the timed region replays, mean of %d launches (min %.2f, max %.2f us)"
               % (runner.spg, len(durs), min(durs), max(durs)))
    elif getattr(eng, "chain", False) and eng._chain_input_ok(bt0):
        # the steps run exactly as in the timed region: a (pipelined) hipGraph of 4 train steps on 4 resident batches, captured
        # with the stamp buffer attached; after each replay the buffer holds the stamps of the graph's LAST step
        from wide_deep_amd import pipeline
        nt = (bt0.B + eng.chain_rt - 1) // eng.chain_rt
        ts = torch.zeros(2 * nt, dtype=torch.int64, device="cuda")
        eng._chain_tile_stamps = ts.data_ptr()
        spans, spreads = [], []
        try:
            nb = len(dev_batches)
            graphs = [pipeline.StepGraph(eng, [dev_batches[(j + i) % nb] for i in range(4)], stream=side)
                      for j in range(0, min(nb, 16), 4)]
            for i in range(steps + 2):
                graphs[i % len(graphs)].replay()
                torch.cuda.synchronize()
                v = ts.cpu().view(nt, 2)
                if i >= 2:
                    spans.append(float(v[:, 1].max() - v[:, 0].min()) / 100.0)       # us
                    spreads.append(float(v[:, 0].max() - v[:, 0].min()) / 100.0)
            del graphs
        finally:
            eng._chain_tile_stamps = None
        us = sum(spans) / len(spans)
        kernel = "k_tower_chain<%d> (input-layer phase)" % eng.chain_rt
        how = ("realtime-clock stamps of all %d workgroups, first start -> last x tile complete, in the last step of a 4-step "
               "hipGraph replay (as timed), mean of %d replays; the workgroups' start skew of %.2f us is included"
               % (nt, len(spans), sum(spreads) / len(spreads)))
    else:
        st = side.cuda_stream
        with torch.cuda.stream(side):
            bts = [synth_hash(eng, tb) for tb in dev_batches]
            for i in range(5):
                eng._sparse_forward(bts[i % len(bts)], st)
            side.synchronize()
            us = event_time_ms(lambda i: eng._sparse_forward(bts[i % len(bts)], st), 100) * 1e3
        fused = eng.spec.has_deep and eng._fused_input_layer and eng.rec is None and not eng._small_on(bt0)
        kernel = ("k_input_layer<%d, 2, %s>" % (dim // 4, "true" if bt0.one_hot else "false") if fused else
                  "the input layer as the step launches it: %s + k_wide_fwd + k_dense_fwd%s"
                  % ("k_embag_fwd<%d> on row records" % (dim // 4) if eng.rec is not None else "k_embag_fwd_range per embedding width",
                     " + k_small_fwd (crossed columns, tables in LDS)" if eng._small_on(bt0) else ""))
        how = "HIP events around the launch%s (in the step as timed here), 100 times over the resident pool" % ("" if fused else "es")
    gbs = alg / us / 1e3
    extra = {}
    rp = rocprof_instep_record() if (span is not None and runner is not None and runner.multis) else None
    if rp and rp.get("in_step_us_mean"):
        # `frac` is the stamp figure (first workgroup's first instruction -> last workgroup's stores landed, measured live in the
        # timed graphs); the kernel-trace duration of the same launch also holds the dispatch ramp in front of the first wavefront
        # and the end-of-kernel release behind the last one
        extra = {"rocprof_instep_us": rp["in_step_us_mean"],
                 "frac_rocprof_instep": round(alg / rp["in_step_us_mean"] / 1e3 / HBM_PEAK_GBS, 4),
                 "rocprof_source": "profiles/r6_gather_instep_rocprof.json (rocprofv3 --kernel-trace of this command, %d in-step "
                                   "launches; another box than this run's)" % rp.get("in_step_launches", 0)}
    return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), **extra, "traffic": None,
            "traffic_source": "PMC counters are per launch, not per overlap window: the traffic of this launch, measured alone, is "
                              "in roofline_gather_kernel",
            "algorithmic_bytes_per_launch": int(alg), "avg_launch_us": round(us, 2), "measured": how,
            "note": "algorithmic bytes = SURVEY 8(d) gather contract only; the launch also reads the wide weight of every occurrence "
                    + ("(row records: it sits in the line fetched for the embedding row)" if eng.rec is not None
                       else "(one more 16-byte {w,z,n} line per occurrence)")
                    + ", the numeric columns, and writes the wide weights / logit (not counted)"
                    + ("; it shares the chip with the tower kernel of the previous batch, off the critical path of the step"
                       if getattr(eng, "prefetch", False) else "")}


def synth_hash(eng, tb):
    from wide_deep_amd import synth
    return synth.hash_tokens(eng, tb)


MFMA_F32_PEAK_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md)


def tower_roofline(eng, bt, iters=100):
    """The one-launch tower (k_tower_chain, the kernel with the largest share of the step): algorithmic flops of the
    forward products and of the input-gradient chain / measured duration (HIP events on the launch stream)."""
    if not getattr(eng, "chain", False):
        return None
    tw = eng.towers[0]
    st = torch.cuda.current_stream().cuda_stream
    B = bt.B
    fuse = eng._chain_input_ok(bt)        # as launched in the step: the input layer is the kernel's first phase
    run = lambda i: eng._tower_chain(tw, bt, B, st, True, fuse)
    for i in range(5):
        run(i)
    torch.cuda.synchronize()
    ms = event_time_ms(run, iters)
    metas, L = tw["metas"], tw["L"]
    dxc = (tw["dx_cols"] + 31) // 32 * 32 if eng.group_slots else 0
    macs = sum(m["K"] * m["N"] for m in metas[:L]) + sum(m["K"] * m["N"] for m in metas[1:L]) + metas[0]["N"] * dxc
    macs += 2 * metas[L]["K"]                      # logits layer forward + its input gradient
    fl = 2.0 * B * macs
    tf = fl / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "k_tower_chain", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "algorithmic_flops_per_launch": int(fl),
            "avg_launch_us": round(ms * 1e3, 2),
            "note": "the kernel as launched in the step (%s): forward of all hidden layers + logits + head + "
                    "input-gradient chain down to dx; exact fp32 MFMA; flops of the products only"
                    % ("x gathered by wd_prefetch_onehot a step ahead: the kernel reads x from HBM and sums the wide weight list"
                       if (fuse and getattr(eng, "prefetch", False)) else
                       "input layer fused: gather phase included in the duration" if fuse else "x from HBM")}


def cpu_baseline(eng, host_batches, steps, B):
    """The CPU oracle (oracle/, a port restating the reference's TF semantics -- TF itself is not installable
    here) timed on this box's host cores over `steps` batches of the same workload, hashing included."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from oracle.harness import oracle_batch, oracle_from_engine
    from wide_deep_amd import synth
    ncpu = os.cpu_count() or 1
    ora = oracle_from_engine(eng)
    plan = eng.plan
    S = plan.S
    nb = [s.num_buckets for s in plan.slots]

    # tokens packed ahead of the timed region, exactly as the GPU path receives them (resident, packed)
    packed = {}
    for hb in host_batches:
        if "features" in hb:
            continue
        data, offs = synth.pack_decimal_tokens(hb["raw"])
        packed[id(hb)] = (data, offs.astype(np.int64))

    def one(hb):
        if "features" in hb:      # parsed batch (crossed columns): fingerprints + SparseCross of the oracle, then the step
            from oracle.harness import parsed_batch_ids
            ids, bag_offs = parsed_batch_ids(plan, hb)
        else:
            lens = hb["lens"]
            slot_of = np.repeat(np.tile(np.arange(S), hb["B"]), lens.reshape(-1))
            data, offs = packed[id(hb)]
            fp = O.fingerprint64_batch(data, offs)
            ids = (fp % np.asarray(nb, dtype=np.uint64)[slot_of]).astype(np.int64)
            bag_offs = synth.offsets_from_lens(lens)
        w = None
        if eng.spec.use_weight_column:
            w = np.where(hb["labels"] > 0, eng.spec.pos_weight, eng.spec.neg_weight).astype(np.float32)
        ob = oracle_batch(plan, ids, bag_offs, hb["B"], hb["dense"], hb["labels"], w)
        return ora.train_step(ob)

    ora.batched = True      # all sparse columns of a step in one C call each way, columns side by side under OpenMP
    one(host_batches[0])    # warm-up (page-in)

    def timed(nt, n):
        torch.set_num_threads(nt)
        O.lib().wdo_set_num_threads(nt)
        one(host_batches[0])
        t0 = time.perf_counter()
        for i in range(n):
            one(host_batches[(i + 1) % len(host_batches)])
        return time.perf_counter() - t0

    # >= 50 timed steps, one thread AND all cores; a mid-size pool is timed too because the port's sparse ops
    # are short loops (more threads is not always faster) -- `value` is the best multi-thread figure, `cores` its threads
    runs = {}
    for nt in sorted({1, min(ncpu, 32), ncpu}):
        n = steps                                          # >= 50 timed steps on every leg (one thread: ~8 s)
        dt = timed(nt, n)
        runs[nt] = {"threads": nt, "steps": n, "seconds": round(dt, 2), "examples_per_sec": round(n * B / dt, 1)}
    multi = [r for nt, r in runs.items() if nt > 1] or list(runs.values())
    best = max(multi, key=lambda r: r["examples_per_sec"])
    return {"value": best["examples_per_sec"], "unit": "examples/sec", "cores": best["threads"], "host_cpus": ncpu,
            "kind": "port", "one_thread": runs[1], "all_cores": runs[ncpu], "runs": list(runs.values()),
            "sample": "%d steps x batch %d of the same synthetic workload (token hashing + full train step) per thread "
                      "count; CPU oracle = C/OpenMP sparse ops (columns side by side) + torch-CPU fp32 tower; TensorFlow "
                      "(the reference's runtime) is not installable in this image.  `value` is the BEST thread count (%d "
                      "threads); the port does not scale past it: all %d host cores give %.0f examples/s (%.2f x), one thread %.0f"
                      % (steps, B, best["threads"], ncpu, runs[ncpu]["examples_per_sec"],
                         runs[ncpu]["examples_per_sec"] / best["examples_per_sec"], runs[1]["examples_per_sec"]),
            "seconds": best["seconds"]}


def cpu_baseline_sharded(spec, args, B):
    """N > 1: the CPU port timed on rank 0's host on the per-rank workload (batch B of the global model), from a child process
    that owns no GPU state of this run: `python bench.py --cpu-only`.  The figure is per-process examples/sec of ONE host, as
    at N = 1 -- the reference's CPU path does not get faster with more GPUs."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-only", "--config", args.config, "--dist", args.dist, "--batch", str(B),
           "--cpu-steps", str(min(args.cpu_steps, 20))]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC") and not k.startswith("WD_FAULT")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"value": None, "error": str(e)[:200]}


def parity_check(eng, spec, tb, hb, step_eager):
    """One eager step on batch 0 BEFORE anything is timed, against the CPU oracle (the checker; sampled-row tables,
    oracle/harness.CompactOracle): hashed ids bit-exact, loss and max |logit - oracle logit| of the step."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from oracle.harness import CompactOracle
    from wide_deep_amd import synth
    plan = eng.plan
    bt = synth.hash_tokens(eng, tb)
    torch.cuda.synchronize()
    ids, offs = bt.ids.cpu().numpy()[: bt.nnz].copy(), bt.bag_offs.cpu().numpy()
    if "features" in hb:       # the device featurizer's ids (hash slots + crossed columns) against the oracle's
        from oracle.harness import parsed_batch_ids
        want, woffs = parsed_batch_ids(plan, hb)
        ids_ok = bool(np.array_equal(offs[: len(woffs)], woffs) and np.array_equal(ids.astype(np.int64), want))
    else:
        data, toffs = synth.pack_decimal_tokens(hb["raw"])
        nb = np.asarray([s.num_buckets for s in plan.slots], dtype=np.uint64)
        slot_of = np.repeat(np.tile(np.arange(plan.S), hb["B"]), hb["lens"].reshape(-1))
        want = (O.fingerprint64_batch(data, toffs.astype(np.int64)) % nb[slot_of]).astype(np.int64)
        ids_ok = bool(np.array_equal(ids.astype(np.int64), want))
    co = CompactOracle(eng, [(ids, offs, hb["B"])])
    w = None
    if spec.use_weight_column:
        w = np.where(hb["labels"] > 0, spec.pos_weight, spec.neg_weight).astype(np.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        loss = float(step_eager(tb))
    torch.cuda.synchronize()
    oloss, ologits = co.ora.train_step(co.batch(ids, offs, hb["B"], hb["dense"], hb["labels"], w))
    d = (eng.logit[: hb["B"]].cpu() - ologits).abs()
    rel = d / (1.0 + ologits.abs())
    return {"batch": 0, "hash_ids_bit_exact": ids_ok, "loss": round(loss, 4), "oracle_loss": round(float(oloss), 4),
            "max_abs_dlogit": float("%.3g" % float(d.max())), "max_rel_dlogit": float("%.3g" % float(rel.max())),
            "tolerance": "fp32 tower: |dlogit| <= 2e-4 + 2e-4 |logit| (tests/test_gpu_fullsize.py); fp16-operand tower 3e-2 + 2e-2 |logit|",
            "oracle": "oracle/ CPU restatement on the rows this batch touches (oracle/harness.CompactOracle)"}


def parity_check_sharded(eng, spec, tb, hb, step_eager, rank):
    """N > 1 (COLLECTIVE: every rank calls it, rank 0 gets the object): one eager sharded step on every rank's batch 0 BEFORE
    anything is timed; rank 0's LOCAL examples against the CPU oracle -- hashed (global) ids bit-exact, logits and the local
    loss sum of the step's forward.  The oracle's tables hold the rows rank 0's batch touches, fetched from their owners by
    plain indexing + a reduce (oracle/harness.ShardedCompactOracle), not through the engine's exchange kernels; the dense
    parameters are the replicated ones.  (The backward / update of N ranks against one GPU on the global batch:
    tests/test_gpu_dist.py, world 2 and 4.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from oracle.harness import ShardedCompactOracle
    from wide_deep_amd import synth
    plan = eng.hash_plan
    bt = synth.hash_tokens(eng, tb)
    torch.cuda.synchronize()
    ids, offs = bt.ids.cpu().numpy()[: bt.nnz].copy(), bt.bag_offs.cpu().numpy()
    ids_ok = None
    if rank == 0 and "features" in hb:      # parsed batch: the device featurizer's ids (hash slots + crossed columns)
        from oracle.harness import parsed_batch_ids
        want, woffs = parsed_batch_ids(plan, hb)
        ids_ok = bool(np.array_equal(offs[: len(woffs)], woffs) and np.array_equal(ids.astype(np.int64), want))
    elif rank == 0:
        data, toffs = synth.pack_decimal_tokens(hb["raw"])
        nb = np.asarray([s.num_buckets for s in plan.slots], dtype=np.uint64)
        slot_of = np.repeat(np.tile(np.arange(plan.S), hb["B"]), hb["lens"].reshape(-1))
        want = (O.fingerprint64_batch(data, toffs.astype(np.int64)) % nb[slot_of]).astype(np.int64)
        ids_ok = bool(np.array_equal(ids.astype(np.int64), want))
    co = ShardedCompactOracle(eng, [(ids, offs, hb["B"])], dst=0)
    w = None
    if spec.use_weight_column:
        w = np.where(hb["labels"] > 0, spec.pos_weight, spec.neg_weight).astype(np.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step_eager(tb)
    torch.cuda.synchronize()
    if rank != 0:
        return None
    B = hb["B"]
    oloss, ologits = co.ora.train_step(co.batch(ids, offs, B, hb["dense"], hb["labels"], w))
    d = (eng.logit[:B].cpu() - ologits).abs()
    rel = d / (1.0 + ologits.abs())
    # the local loss sum from the logits the engine produced (the engine's own `loss` may already be the all-reduced one)
    x = eng.logit[:B].cpu().double()
    y = torch.as_tensor(hb["labels"], dtype=torch.float64)
    ww = torch.as_tensor(w, dtype=torch.float64) if w is not None else torch.ones(B, dtype=torch.float64)
    loss = float((ww * (x.clamp_min(0) - x * y + torch.log1p(torch.exp(-x.abs())))).sum())
    return {"batch": 0, "rank": 0, "world": eng.world, "what": "rank 0's local examples of one eager sharded step (forward)",
            "hash_ids_bit_exact": ids_ok, "loss_local": round(loss, 4), "oracle_loss_local": round(float(oloss), 4),
            "max_abs_dlogit": float("%.3g" % float(d.max())), "max_rel_dlogit": float("%.3g" % float(rel.max())),
            "tolerance": "|dlogit| <= 2e-4 + 2e-4 |logit| (tests/test_gpu_fullsize.py)",
            "oracle": "oracle/ CPU restatement on the rows rank 0's batch touches, fetched from their owners by indexing + "
                      "reduce (oracle/harness.ShardedCompactOracle)"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec the same command line as N ranks of ONE node
    (python -m torch.distributed.run, rendezvous on 127.0.0.1 and a free port) and wait for them."""
    import socket
    import subprocess
    backend = os.environ.get("WD_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        raise SystemExit("bench.py --gpus %d: this node has %d GPU(s); RCCL needs one device per rank "
                         "(WD_DIST_BACKEND=gloo stages the collectives through the host with ranks sharing a GPU: "
                         "a functional check, not a measurement)" % (n, have))
    def launch(extra_env):
        with socket.socket() as s2:
            s2.bind(("127.0.0.1", 0))
            p = s2.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(p), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        env.update(extra_env)
        return subprocess.call(cmd, env=env)

    rc = launch({})
    if rc != 0 and "WD_DIST_GRAPH" not in os.environ:
        # The ranks died (capturing RCCL collectives into a hipGraph SEGFAULTS hipStreamEndCapture of ROCm 7.2 when one of its
        # capture rules is broken -- DESIGN.md section 6 -- and a signal cannot be caught in-process) or left with an error:
        # once more with hipGraph segments between ORDINARY collectives, so that the run still ends with a measured line
        # (`config.exchange.graph_fallback` records why)
        print("bench: the %d ranks ended with status %d; retrying with WD_DIST_GRAPH=segments" % (n, rc), file=sys.stderr, flush=True)
        rc = launch({"WD_DIST_GRAPH": "segments", "WD_BENCH_GRAPH_FALLBACK": "first attempt (collectives captured into the "
                     "step graphs) ended with status %d" % rc})
    raise SystemExit(rc)


def probe_capture_child():
    """`bench.py --probe-capture` (started by probe_captured_collectives, one child per rank, on a rendezvous of their own):
    a small C2-shaped model through exactly what the timed region of an N > 1 run is made of -- dist.ShardedStepGraph, the
    all-to-alls and the all-reduce captured into multi-step hipGraphs, built by pipeline.StepRunner -- captured, replayed,
    segment overflow checked.  Leaves with os._exit: 0 = it works on this node."""
    import datetime
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=datetime.timedelta(minutes=2))
    from wide_deep_amd import pipeline, synth
    from wide_deep_amd.dist import ShardedStepGraph, ShardedWideDeepEngine
    from wide_deep_amd.plan import criteo_spec
    B = 1024
    spec = criteo_spec(n_dense=13, n_sparse=26, buckets=20000, dim=16, hidden=(256, 128, 64))
    eng = ShardedWideDeepEngine(spec, max_batch=B, seed=0, slack=2.0)
    tbs = [synth.TokenBatch(eng.hash_plan, synth.make_raw_batch(eng.hash_plan, B, seed=77 + 100 * rank + i)) for i in range(4)]
    if not (eng._graph_mode() == "full" and eng.chain and eng._chain_input_ok(tbs[0].batch)):
        print("probe: this model does not take the captured-collectives path", flush=True)
        os._exit(3)
    if os.environ.get("WD_FAULT_CAPTURE_RANK") == str(rank):     # test hook: die the way a refused capture does
        import signal
        os.kill(os.getpid(), signal.SIGSEGV)
    runner = pipeline.StepRunner(eng, tbs, 4, 2, False, ShardedStepGraph, n_singles=2)
    runner.warm_up(1)
    runner.run(5)
    torch.cuda.synchronize()
    eng.check_overflow()
    ok = bool(torch.isfinite(eng.loss).all())
    print("probe ok" if ok else "probe: loss is not finite", flush=True)
    os._exit(0 if ok else 4)       # (no destroy_process_group: it can hang behind captured collectives)


def probe_captured_collectives(rank, world, limit=240):
    """Can this node capture RCCL collectives into hipGraphs?  Tried in a CHILD process per rank before anything is built.
    hipStreamEndCapture of ROCm 7.2 answers a broken capture rule with SIGSEGV: under
    `python -m torch.distributed.run` -- how the driver starts an N > 1 run; self_launch() can retry, that launcher cannot --
    one rank dying takes the job with it and no line is printed.  The children rendezvous on a port of their own; every rank
    waits for its child (a hang counts as a failure after `limit` seconds), the ranks agree (MIN), and on a failure the run
    uses hipGraph segments between ordinary collectives.  Returns None (captured collectives work) or why not."""
    import socket
    import subprocess
    port = torch.zeros(1, dtype=torch.int32, device="cuda")
    if rank == 0:
        with socket.socket() as s2:
            s2.bind(("127.0.0.1", 0))
            port[0] = s2.getsockname()[1]
    if world > 1:
        torch.distributed.all_reduce(port)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}     # rank 0's child hosts the store
    env["MASTER_PORT"] = str(int(port.item()))
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    why = None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-capture"], env=env, timeout=limit,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0 or b"probe ok" not in r.stdout:
            tail = r.stdout.decode("utf-8", "replace").strip().splitlines()[-1:] or [""]
            why = "status %d (%s)" % (r.returncode, tail[0][:120])
    except subprocess.TimeoutExpired:
        why = "no answer in %d s" % limit
    flag = torch.tensor([0 if why else 1], dtype=torch.int32, device="cuda")
    if world > 1:
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    if int(flag.item()):
        return None
    return "probe of captured collectives failed on %s" % ("this rank: " + why if why else "another rank")


def cpu_only(args):
    """`cpu_baseline` of the workload and nothing else (one JSON line): a single-GPU engine supplies the model's initial state,
    the oracle is timed on the host."""
    from wide_deep_amd import synth
    from wide_deep_amd.engine import WideDeepEngine
    torch.cuda.set_device(0)
    spec, mean_len = make_spec(args.config)
    B = args.batch
    eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * (2 * mean_len + 2), seed=0)
    if any(s.kind == "cross" for s in spec.slots):
        hbs = [synth.make_parsed_batch(eng.plan, B, seed=20260925 + i, mean_len=mean_len, dist=args.dist,
                                       weights=(spec.pos_weight, spec.neg_weight))[1] for i in range(4)]
    else:
        hbs = [synth.make_raw_batch(eng.plan, B, seed=20260925 + i, mean_len=mean_len, dist=args.dist) for i in range(4)]
    print(json.dumps(cpu_baseline(eng, hbs, args.cpu_steps, B)), flush=True)


def main():
    args = parse()
    if os.environ.get("WD_HANG_DUMP"):      # diagnostics: dump every thread's stack and exit if the run takes this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["WD_HANG_DUMP"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.cpu_only:
        return cpu_only(args)
    if args.probe_capture:
        return probe_capture_child()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- one rank per GPU through torch.distributed.run on a free
        # local port; rank 0 prints the one JSON line, this process forwards the ranks' exit status
        return self_launch(args.gpus)
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d, but WORLD_SIZE=%d: launch with python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d ... (or plain `python bench.py --gpus %d`)"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    # WD_DIST_BACKEND=gloo: functional smoke test of the N>1 path with all ranks on ONE GPU (host-staged collectives)
    backend = os.environ.get("WD_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:     # diagnostics: the sharded step with a one-rank RCCL group (exchange kernels + collectives, no peers)
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # a rank that loses its peers fails after 5 minutes instead of holding the node for the default 10+
        import datetime
        limit = datetime.timedelta(minutes=5)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
        if (backend == "nccl" and not args.no_graph and "WD_DIST_GRAPH" not in os.environ and args.config in ("c2", "c3")
                and (world > 1 or os.environ.get("WD_BENCH_PROBE") == "1") and os.environ.get("WD_BENCH_PROBE") != "0"):
            why = probe_captured_collectives(rank, world)
            if why:
                print("bench: %s; hipGraph segments between ordinary collectives" % why, file=sys.stderr, flush=True)
                os.environ["WD_DIST_GRAPH"] = "segments"
                os.environ.setdefault("WD_BENCH_GRAPH_FALLBACK", why)

    from wide_deep_amd import pipeline, synth
    from wide_deep_amd.engine import WideDeepEngine

    spec, mean_len = make_spec(args.config)
    featurized = any(s.kind == "cross" for s in spec.slots)
    tower_dtype = args.tower_dtype or ("fp16" if args.config == "c5" else "fp32")
    B = args.batch
    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit("--scaling strong: --batch %d (global) must be a multiple of the %d ranks" % (args.batch, world))
        B = args.batch // world
    parsed = None
    if featurized:
        # crossed columns (BASELINE configs[3]): the batches are PARSED batches resident in HBM (token bytes + per-feature example
        # ranges, what dataset.input_fn hands over) and every step runs the product's device featurizer on its batch
        # (features.Featurizer.run: fingerprints, bag lengths, bag CSR, hash buckets, SparseCross -- no host wait) like C2's steps
        # run wd_hash_bucket; --ids-input featurizes before the timed region instead (the line of round 5)
        from wide_deep_amd.plan import FeaturePlan
        gp = FeaturePlan(spec)
        parsed = [synth.make_parsed_batch(gp, B, seed=20260925 + 1000 * rank + i, mean_len=mean_len, dist=args.dist,
                                          weights=(spec.pos_weight, spec.neg_weight) if spec.use_weight_column else None)
                  for i in range(args.pool)]
    if sharded:
        from wide_deep_amd.dist import ShardedWideDeepEngine
        # distinct (slot, id) pairs of this rank's first batch: what the exchange segments are sized from when the engine sends
        # every distinct row once (sender-side unique; WD_SHARD_DEDUP=auto turns it on below 80 % distinct)
        from wide_deep_amd.plan import FeaturePlan
        uniq = None
        if mean_len == 1 and os.environ.get("WD_SHARD_DEDUP", "auto") != "0":
            gp = FeaturePlan(spec)
            r0 = synth.make_raw_batch(gp, B, seed=20260925 + 1000 * rank, mean_len=1, dist=args.dist)["raw"].reshape(B, -1)
            uniq = sum(len(np.unique(r0[:, j])) for j in range(r0.shape[1]))
        dedup_expected = uniq is not None and (uniq < 0.8 * B * 26 or os.environ.get("WD_SHARD_DEDUP") == "1")
        eng = ShardedWideDeepEngine(spec, max_batch=B, seed=0,
                                    max_nnz=(int(1.02 * max(hb["nnz"] for _, hb in parsed)) + 1024) if featurized else B * 26 * (2 * mean_len + 2),
                                    expected_nnz=B * 26 * mean_len, expected_unique=uniq,
                                    # per-peer segment capacity over the uniform expectation: Zipf(1.05) sends ~8 % of all
                                    # occurrences to the owner of the hottest row (id % world), 1.6x the mean at 8 ranks --
                                    # but only one REQUEST when the row travels once
                                    # Uniform ids / distinct rows: a peer receives Binomial(n, 1/W) entries -- at W = 8,
                                    # n = 213 k that is 26.6 k +- 152, so 1.1 leaves 17 sigma (round 3 shipped 1.3: 15 % more
                                    # bytes in every all-to-all and in the owner-side kernels that walk the padding)
                                    slack=args.slack if args.slack else (1.1 if args.dist == "uniform" else 1.2 if dedup_expected else 2.5))
    elif featurized:
        eng = WideDeepEngine(spec, max_batch=B, max_nnz=int(1.02 * max(hb["nnz"] for _, hb in parsed)) + 1024, seed=0,
                             tower_dtype=tower_dtype, expected_nnz=B * 28 * mean_len)
    else:
        # (multi-hot: the bucket geometry is sized from the occurrences a batch really holds -- ~130 per bucket, what the bucket
        # sort's 256-pair workgroups take -- not from one id per bag)
        eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * (2 * mean_len + 2), seed=0, tower_dtype=tower_dtype,
                             expected_nnz=B * 26 * mean_len if mean_len > 1 else None)
    plan = getattr(eng, "hash_plan", eng.plan)    # sharded: batches live in the GLOBAL id space

    # resident batch pool (raw tokens in HBM); distinct seeds per rank
    host_batches, dev_batches = [], []
    if featurized:
        from wide_deep_amd.features import Featurizer
        fz = Featurizer(eng, cross_padding="ragged")
        for i, (raw, hb) in enumerate(parsed):
            if i < 4:
                host_batches.append(hb)
            dev_batches.append(synth.FeaturizedBatch(fz.to_device(raw), hb) if args.ids_input
                               else synth.ParsedTokenBatch(fz, raw, hb, ids_capacity=eng.max_nnz))
        del parsed
    for i in range(0 if featurized else args.pool):
        hb = synth.make_raw_batch(plan, B, seed=20260925 + 1000 * rank + i, mean_len=mean_len, dist=args.dist)
        w = None
        if spec.use_weight_column:
            w = np.where(hb["labels"] > 0, spec.pos_weight, spec.neg_weight).astype(np.float32)
        if i < 4:
            host_batches.append(hb)
        dev_batches.append(synth.TokenBatch(plan, hb, weights=w))
    if args.ids_input and not featurized:
        for tb in dev_batches:
            synth.hash_tokens(eng, tb)
    torch.cuda.synchronize()

    def step_eager(tb):
        return pipeline.step_eager(eng, tb, args.ids_input)

    parity = None
    if not sharded and not args.no_parity:
        parity = parity_check(eng, spec, dev_batches[0], host_batches[0], step_eager)
    elif sharded and not args.no_parity:
        parity = parity_check_sharded(eng, spec, dev_batches[0], host_batches[0], step_eager, rank)      # collective

    use_graph = not args.no_graph
    steps_per_run, run_steps, run_steps_graph, chain = 1, None, False, False
    graph_cls = pipeline.StepGraph
    if use_graph and sharded and eng._graph_mode() == "full" and eng.chain and eng._chain_input_ok(dev_batches[0].batch):
        # RCCL backend: the collectives are captured with the kernels -> multi-step pipelined graphs like the single-GPU path
        from wide_deep_amd.dist import ShardedStepGraph
        graph_cls = ShardedStepGraph
    multis, runner = [], None
    if sharded and os.environ.get("WD_FAULT_CAPTURE_RANK") == str(rank) and "WD_BENCH_GRAPH_FALLBACK" not in os.environ:
        # test hook (tests/test_gpu_dist.py): this rank dies the way a refused capture does -- by signal, uncatchable
        import signal
        sys.stderr.write("bench: fault injection: rank %d dies at graph capture\n" % rank)
        sys.stderr.flush()
        os.kill(os.getpid(), signal.SIGSEGV)
    span = None
    if use_graph and not sharded and getattr(eng, "prefetch", False) and eng._chain_input_ok(dev_batches[0].batch):
        # diagnostics that ride in the timed graphs: start / end stamps of every workgroup of wd_prefetch_onehot (two 8-byte
        # stores per workgroup), read after the timed region for `roofline`
        span = torch.zeros(eng.n_act, eng.prefetch_blocks(B), 2, dtype=torch.int64, device="cuda")
        eng._prefetch_span = span.data_ptr()

    def build_runner():
        """pipeline.StepRunner: the multi-step (chained) hipGraphs over windows of the resident pool -- the object that is timed,
        and the one tests/test_gpu_fullsize.py replays against the oracle.  Captures only; warm_up() replays."""
        return pipeline.StepRunner(eng, dev_batches, args.steps, args.steps_per_graph, args.ids_input, graph_cls,
                                   n_singles=8 if sharded else None)

    if use_graph and graph_cls is not pipeline.StepGraph:
        # captured collectives: if ANY rank's runtime refuses the capture, every rank falls back to graph segments between
        # ordinary collectives (the ranks must keep issuing the same collectives in the same order)
        ok, err = 1, None
        try:
            runner = build_runner()          # captures only: nothing has been replayed when the ranks compare notes
        except Exception as e:
            ok, err = 0, e
        if world > 1:
            flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            runner.warm_up()
            run_steps, steps_per_run, chain, multis, run_steps_graph = runner.run, runner.spg, runner.chain, runner.multis, True
        else:
            runner = None
            print("bench: multi-step graphs with captured collectives unavailable (%s); graph segments between the collectives"
                  % (err if err is not None else "refused on another rank"), file=sys.stderr)
            torch.cuda.synchronize()
            os.environ["WD_DIST_GRAPH"] = "segments"
            graph_cls, run_steps, run_steps_graph, steps_per_run, chain, multis = pipeline.StepGraph, None, False, 1, False, []
    if use_graph and sharded and graph_cls is pipeline.StepGraph:
        # sharded step: hipGraph segments between the collectives (dist._Segments); every rank captures in lock-step
        try:
            # (one eager warm-up step in front of the first capture only: the batches share every buffer shape)
            replays = [eng.capture_train_step(tb.batch, warmup=1 if i == 0 else 0,
                                              pre=None if args.ids_input else (lambda tb=tb: synth.hash_tokens(eng, tb)))
                       for i, tb in enumerate(dev_batches)]
            run = lambda i: replays[i % len(replays)]()
        except Exception as e:      # capture refused by the runtime: the eager step is the same work, launch by launch
            print("bench: graph segments unavailable (%s); running the sharded step eagerly" % (e,), file=sys.stderr)
            torch.cuda.synchronize()
            use_graph = False
            run = lambda i: step_eager(dev_batches[i % len(dev_batches)])
    elif use_graph and graph_cls is pipeline.StepGraph:
        runner = build_runner()
        runner.warm_up()
        run_steps, steps_per_run, chain, multis, run_steps_graph = runner.run, runner.spg, runner.chain, runner.multis, True
    else:
        run = lambda i: step_eager(dev_batches[i % len(dev_batches)])
    if run_steps is None:
        def run_steps(n):
            for i in range(n):
                run(i)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    # the K timed steps, `--repeats` times back to back; every repeat is bracketed by barrier + synchronize and reduced with MAX
    # over the ranks.  `value` is the MEDIAN repeat: one stalled lease of the box (seen once in ~35 runs) cannot halve it.
    reps = []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        reps.append(el)
    elapsed = sorted(reps)[len(reps) // 2]
    loss = float(eng.loss)
    overflow = None
    if sharded:
        try:
            eng.check_overflow()     # collective; one device->host read AFTER the timed region
            overflow = "clean"
        except Exception as e:
            overflow = str(e)[:160]
    value = args.steps * B * world / elapsed

    out = {
        "metric": "examples/sec", "value": round(value, 1), "unit": "examples/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "repeats_ms_per_step": [round(r / args.steps * 1e3, 4) for r in reps],
        "ms_per_step_min_median_max": [round(x / args.steps * 1e3, 4) for x in (min(reps), elapsed, max(reps))],
        "dtype": "f32" if tower_dtype == "fp32" else "f16 tower operands, f32 accumulate / embeddings / optimizer state",
        "data": "synthetic",
        "config": {
            "workload": {
                "c2": "BASELINE configs[1] (C2): Criteo-shape synthetic, 13 dense + 26 sparse slots x 1M hash buckets, "
                      "emb_dim 16, Dnn [256,128,64] BN+ReLU, wide FTRL + deep Adagrad, batch %d per GPU" % B,
                "c3": "BASELINE configs[2] (C3) table: C2 with ONE 100M-row table (26 x 3,846,154 rows) on %d GPU(s), "
                      "batch %d per GPU" % (world, B),
                "c4": "BASELINE configs[3] (C4) on one GPU: multi-hot avg %d ids/slot (1 + Poisson(4), <= 32), 26 slots x 1M buckets + "
                      "200-bucket crossed columns over slots (0,1) and (2,3,4) (ragged products: ~25 and ~125 ids per example), "
                      "ResDnn [256,128,64], weight column 0.99 / 0.01, batch %d" % (mean_len, B),
                "c4-nocross": "BASELINE configs[3] shape WITHOUT its crossed columns (the workload rounds 1-4 reported as C4) on %d "
                              "GPU(s): multi-hot avg %d ids/slot, ResDnn, weight column" % (world, mean_len),
                "c5": "BASELINE configs[4] (C5): deep-only DenseDnn [1024,512,256,128], emb_dim 64, %s tower, batch %d"
                      % (tower_dtype, B),
            }[args.config],
            "global_batch": B * world, "ids": args.dist,
            "input": ("ids resident: produced before the timed region by the device featurizer (hash buckets + crossed columns)"
                      if featurized and args.ids_input else
                      "parsed batch resident (token bytes + example ranges); fingerprints, bag CSR, hash buckets and crossed columns in step"
                      if featurized else "pre-hashed ids" if args.ids_input else "raw string tokens (hashed in step)"),
            # machine-readable: is the tokens -> ids work (hash buckets, crossed columns) inside the timed region?
            "featurize_in_timed_region": not args.ids_input,
            "hip_graph": bool(use_graph), "steps_per_graph": steps_per_run,
            "chained_graphs": bool(use_graph and run_steps_graph and chain),
            "table_layout": ("row records: %d B = [emb %d f32 | w z n -]" % (4 * eng.rec_stride, eng.emb.shape[1])
                             if getattr(eng, "rec", None) is not None else "separate tables"),
            "pipelined_graph": bool(use_graph and run_steps_graph and (sharded or multis[0].pipelined)),
            "parallelism": "dp%d+row-sharded tables" % world if sharded else "single GPU",
            "final_loss_sum": round(loss, 3),
        },
    }
    if sharded:
        RS, cap, W = eng.RS, eng.cap, world
        out["config"]["exchange"] = {
            "graph": eng._graph_mode() if use_graph else "eager",
            "segment_capacity": cap, "check_overflow": overflow, "sender_side_unique": bool(getattr(eng, "dedup", False)),
            "payload_bytes_per_rank_per_step": {"A_rows_int32": 4 * W * cap, "B_records_f32": 4 * W * cap * RS,
                                                "C_gradients_f32": 4 * W * cap * RS, "D_dense_allreduce_f32": 4 * eng.G.numel()},
            "owner_table_layout": "row records" if eng.rec is not None else "separate tables"}
        if os.environ.get("WD_BENCH_GRAPH_FALLBACK"):
            out["config"]["exchange"]["graph_fallback"] = os.environ["WD_BENCH_GRAPH_FALLBACK"]
        # every collective of the step alone, against the xGMI peak (collective: all ranks measure, rank 0 reports)
        XGMI_PEAK_GBS = 7 * 153.6           # MI355X: 7 links x 153.6 GB/s per direction to the 7 peers
        legs = eng.measure_exchange()
        nreq = int((eng.recv_rows >= 0).sum().item())      # requests this rank answered in the last step
        out["config"]["exchange"]["collectives_alone"] = {
            k: {"us": round(us, 2), "wire_bytes_per_rank": int(nb), "achieved_GBps": round(nb / us / 1e3, 1),
                "frac_of_xgmi_peak": round(nb / us / 1e3 / XGMI_PEAK_GBS, 4)}
            for k, (us, nb) in legs.items() if k != "owner_gather_kernel"}
        out["config"]["exchange"]["xgmi_peak_GBps_per_gpu"] = XGMI_PEAK_GBS
        out["config"]["exchange"]["note"] = ("each collective issued alone, 20 calls between two stream events, full static "
                                             "segments; backend %s%s" % (backend, "" if world > 1 else
                                                                         " (one rank: RCCL's local copies, no link)"))
        us_g = legs["owner_gather_kernel"][0]
        D_ = eng.dim
        alg = nreq * (D_ * 4 + 4) + nreq * eng.RS * 4      # SURVEY 8(d) per request: row read + id, record written back
        out["roofline_sharded_gather"] = {
            "bound": "hbm", "kernel": "wd_owner_gather_rec (this rank's share of the embedding gather: one %d-byte record per "
                                      "received request)" % (4 * eng.rec_stride) if eng.rec is not None else "wd_owner_gather",
            "achieved": round(alg / us_g / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / us_g / 1e3 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(alg),
            "avg_launch_us": round(us_g, 2), "requests": nreq,
            "note": "rank 0, the launch alone on the request list of the last step (HIP events, 20 launches); segments are "
                    "static, so the launch also walks the unused (-1) entries"}
    if rank == 0:
        out["parity"] = parity
        if world == 1:
            # dominant-bandwidth kernel named by the metric: the embedding gather
            for tb in dev_batches:
                synth.hash_tokens(eng, tb)
            torch.cuda.synchronize()
            if not sharded and eng.spec.has_deep and eng.group_slots:
                out["roofline"] = gather_instep_roofline(eng, dev_batches, step_eager, runner=runner, span=span, dump=args.dump_stamps)
            out["roofline_gather_kernel"] = gather_kernel_roofline(eng, [tb.batch for tb in dev_batches], args)
            if "roofline" not in out:
                out["roofline"] = out["roofline_gather_kernel"]
            elif out["roofline"].get("traffic") is None and out["roofline_gather_kernel"].get("traffic") is not None and span is not None:
                # the in-step launch IS this launch (same kernel, same arguments, the same resident batches): its HBM traffic is
                # what the PMC passes measured for it -- counter collection serialises kernels, so the bytes can only be taken
                # with the launch alone; what the overlap changes is its duration, not what it moves
                out["roofline"]["traffic"] = out["roofline_gather_kernel"]["traffic"]
                out["roofline"]["traffic_source"] = ("the same launch under rocprofv3 --pmc (roofline_gather_kernel.traffic_source): "
                                                     "counter passes serialise kernels, so the bytes are those of the launch "
                                                     "alone; inside the step only its duration differs")
            for key in ("roofline", "roofline_gather_kernel"):
                r = out.get(key)
                if r and r.get("traffic") and r.get("avg_launch_us"):
                    # what the launch really moves (128-byte lines, request counters) over its duration: the contract's `frac` is
                    # this figure x algorithmic bytes / traffic
                    r["moved_GBps"] = round(r["traffic"] / r["avg_launch_us"] / 1e3, 1)
                    r["moved_frac_of_peak"] = round(r["traffic"] / r["avg_launch_us"] / 1e3 / HBM_PEAK_GBS, 4)
            if not sharded:
                out["roofline_tower"] = tower_roofline(eng, dev_batches[0].batch)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(eng, host_batches, args.cpu_steps, B)
            else:
                out["cpu_baseline"] = None
        else:
            out["roofline"] = out["roofline_sharded_gather"]
            out["cpu_baseline"] = None
            if not args.no_cpu_baseline:
                # the same CPU port on the same per-rank workload (a single-GPU-shaped oracle of the global model over the rows
                # rank 0's batches touch would be the same arithmetic): a bounded sample, rank 0's host cores while the peers idle
                out["cpu_baseline"] = cpu_baseline_sharded(spec, args, B)
        print(json.dumps(out), flush=True)
    if sharded:
        # graphs that hold captured RCCL nodes go first; the teardown itself runs on a helper thread -- on ROCm 7.2 / RCCL 2.26
        # destroy_process_group() can block forever once collectives have been captured -- and the process leaves regardless
        import threading
        multis = singles = replays = run = run_steps = None
        torch.cuda.synchronize()
        if os.environ.get("WD_DIST_TEARDOWN") == "skip":     # (under rocprofv3: leave through the interpreter's normal exit)
            return
        th = threading.Thread(target=torch.distributed.destroy_process_group, daemon=True)
        th.start()
        th.join(10.0)
        sys.stdout.flush()
        sys.stderr.flush()
        if th.is_alive():
            os._exit(0)


if __name__ == "__main__":
    main()
