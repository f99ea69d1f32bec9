"""Replay of consecutive train steps as ONE hipGraph (the outer loop of python/train.py:65-165 over resident batches).

A train step on a resident batch is a fixed sequence of launches on fixed buffers (engine.py allocates nothing per step),
so `n` consecutive steps -- each on its OWN batch: raw tokens -> wd_hash_bucket -> forward -> loss -> backward -> both
optimizers -- can be captured once and replayed; the ~11 us between two graph launches is then paid once per `n` steps.
Used by bench.py (the timed path) and by tests/test_gpu_fullsize.py (the same path against the CPU oracle).

Pipelined capture (the Criteo-shaped fast path: one-launch tower with the fused input layer, Adagrad + Ftrl).  Captured
from plain stream order, a step is a chain with one side branch, and every step waits for ALL of the previous one
(profiles/r2d_timeline_before_pipelining.txt: hash 4 -> fold 9 -> tower 89 -> weight gradients 39 -> finalize 29 || row update 67, then
6 us of join before the next hash).  The real dependencies are fewer:

    hash(t)                    needs the tokens of batch t only            -> input branch, off the critical path
    bucketing(t)               needs ids(t) (and a scratch set no running update reads)  -> input branch, after tower(t-1)
    tower(t)                   needs ids(t), the rows of update(t-1), the folded weights of dense(t-1)
    products(t) (weight-gradient GEMMs)                                   needs tower(t)
    update(t) (rows: Adagrad / Ftrl)                                      needs tower(t) (dx, dlogit) and bucketing(t)
    tail(t) = split-K sums + Adagrad + packed kernels for t+1            needs products(t)

so the graph is built with these edges (three streams + events during capture): the hash leaves the critical path and the
row update runs beside the MFMA-bound products and the tail (WD_PIPE_TAIL=after joins the update BEFORE the tail, which is
a chain of dependent loads -- 19 us alone, ~40 us beside the update that keeps the memory queues full -- but the extra
cross-queue edge costs more than it saves: 0.198 against 0.185 ms/step, measured in round 2, profiles/README.md).  Every kernel
still runs once per step on the same operands: results are bit-identical to the eager launches
(tests/test_gpu_fullsize.py).  WD_PIPELINE=0 captures plain stream order.  (A fourth kind of edge -- the input branch
waiting for update(t-2) so that bucketing can run a step ahead on a second scratch set -- makes hipStreamEndCapture of
ROCm 7.2 crash from three steps per graph on; bucketing(t) is therefore released by the end of tower(t-1), which update(t-2)
precedes through the dense chain, and alternates between two scratch sets.)
"""
import os

import torch

from . import hipgraph, synth
from .engine import WideDeepEngine


def step_eager(eng, tb, ids_input=False):
    """One train step on a resident TokenBatch: hash in step unless the ids are already there."""
    bt = tb.batch if ids_input else synth.hash_tokens(eng, tb)
    return eng.train_step(bt)


def pipelined_ok(eng, tb):
    """The fast path the pipelined capture knows: single-GPU engine, one-launch tower building its own x tile, the
    reference's default optimizers (fused finalize + Adagrad, fused row update)."""
    bt = tb.batch
    return (type(eng) is WideDeepEngine and eng.spec.has_deep and eng.chain and eng._chain_input_ok(bt)
            and eng.default_opts and eng.all_simple and eng._fold_at_end() and not eng.dropout and eng._has_sparse_update()
            and bt.labels is not None)


class StepGraph:
    """`len(token_batches)` consecutive train steps in one hipGraph.  The engine must have run at least one eager step
    on a side stream before (lazy allocations / module loads are not capturable; that step also leaves the packed kernel
    copies current -- `eng._folded` -- without which the capture is the plain stream-order one, `self.pipelined` False).
    A captured step bakes in "the packed copies are current": `import_state` re-packs them itself; anything else that writes
    the dense parameters behind the engine's back must call `eng._chain_tail(WD_TAIL_PACK)` before the next replay."""

    def __init__(self, eng, token_batches, ids_input=False, stream=None, pipelined=None, lookahead=None, phase=(0, 0),
                 primed=False):
        """`lookahead`, `phase`, `primed` chain consecutive graphs of the prefetched capture (engine.prefetch) the way steps are
        chained inside one: a graph with a `lookahead` batch also hashes, buckets and gathers THAT batch (beside its last
        tower; its last update patches the rows both share), and the next graph -- captured with `primed=True` on that batch
        and with the phase `next_phase` of this one -- starts with its tower.  A primed graph replayed in any other situation
        (`eng._primed` does not name its first batch at the current global step) runs that input work itself first
        (`prime()`), so a chain can be entered anywhere; results are bit-identical either way.  The token is void after anything
        that rewrites a bucket set or an activation buffer outside a replay (an eager `forward()` / train step: the engine
        clears it).  It names the look-ahead batch by identity: a TokenBatch that is refilled IN PLACE between the replay that
        gathered it and the replay that trains on it must be re-primed by the caller (`eng._primed = None`)."""
        self.eng = eng
        if eng.spec.lr_decay:
            import warnings
            warnings.warn("StepGraph: the model decays its learning rates (train.yaml lr_decay), a captured step has the rates of the "
                          "capture baked in -- replays will not decay")
        self.n = len(token_batches)
        self.ids_input = ids_input
        self.first, self.lookahead, self.phase, self.primed = token_batches[0], lookahead, (phase[0], phase[1]), bool(primed)
        self.stream = stream or torch.cuda.Stream()
        self.graph = hipgraph.new_graph()
        if pipelined is None:
            pipelined = os.environ.get("WD_PIPELINE", "1") != "0"
        self.pipelined = bool(pipelined) and all(pipelined_ok(eng, tb) for tb in token_batches) and eng._folded
        self.chained = self.pipelined and eng.prefetch and (lookahead is None or pipelined_ok(eng, lookahead))
        if not self.chained:
            if primed or lookahead is not None or tuple(phase) != (0, 0):
                raise ValueError("StepGraph: lookahead / phase / primed need the prefetched pipelined capture (engine.prefetch)")
        self.stream.wait_stream(torch.cuda.current_stream())
        gs = eng.global_step            # capturing executes nothing: the counter must not move
        with torch.cuda.graph(self.graph, stream=self.stream):
            if self.pipelined and eng.prefetch:
                try:
                    self._capture_prefetched(token_batches, ids_input)
                finally:
                    eng._apar, eng._prefetched = 0, False
            elif self.pipelined:
                self._capture_pipelined(token_batches, ids_input)
            else:
                self._capture_plain(token_batches, ids_input)
        eng.global_step = gs
        self._bump = (3 if eng.spec.model_type == "wide_deep" else 2) * self.n

    def _capture_plain(self, tbs, ids_input):
        """Stream order, step after step.  Batches on the general path behind the one-launch tower (ragged bags:
        eng.lookahead_ok) get two things moved: the hashes of every step to the head of the graph on a branch of their own (24 us
        per step at configs[3] that depend on nothing), and the bucketing of batch t+1 into step t, behind its dense tail and
        beside its row update (engine.train_step)."""
        eng = self.eng
        look = (os.environ.get("WD_LOOKAHEAD", "1") != "0" and hasattr(eng, "lookahead_ok")
                and all(eng.lookahead_ok(tb.batch) and tb.batch.labels is not None for tb in tbs))
        if not look:
            # the tokens -> ids work of batch t+1 (hash buckets; crossed columns: features.Featurizer.run, ~50 us at configs[3]) goes
            # INTO step t: launched behind the step's join with its row update the featurizer sat between two steps, behind the
            # update on the same hardware queue (0.598 -> 0.562 ms at configs[3] from tokens when it moved in front of the join)
            ahead = (not ids_input and type(eng) is WideDeepEngine and os.environ.get("WD_HASH_AHEAD", "1") != "0"
                     and all(tb.batch.labels is not None for tb in tbs))
            if not ahead:
                for tb in tbs:
                    step_eager(eng, tb, ids_input)
                return
            synth.hash_tokens(eng, tbs[0])
            # Where: the HEAD of the next batch's featurizer (fingerprints, bag lengths, bag CSR; a plain token batch: its whole hash)
            # on the side stream between this batch's bucketing and its sort -- beside the input layer and the tower, whose CUs'
            # memory side idles; the EMIT (one lane per id: 64-bit integer VALU, 33 us alone and 120 beside the tower's MFMA stream,
            # where it also ran past the tower's end and held up the row update behind it) on this stream in front of the join with
            # the row update, behind the small tables' update.  WD_HASH_WHERE=join: all of it there (0.510 against 0.498-0.505 ms/step
            # at configs[3]); =tower: all of it on the side stream.
            where = os.environ.get("WD_HASH_WHERE", "split")
            keep = self._events = []
            for t, tb in enumerate(tbs):
                nxt = tbs[t + 1] if t + 1 < len(tbs) else None
                if nxt is None:
                    eng.train_step(tb.batch)
                elif where == "join":
                    eng.train_step(tb.batch, before_join=lambda nxt=nxt: synth.hash_tokens(eng, nxt))
                elif where == "tower":
                    eng.train_step(tb.batch, beside_tower=lambda nxt=nxt: synth.hash_tokens(eng, nxt))
                else:
                    box = {}

                    def head(nxt=nxt, box=box):
                        synth.hash_tokens(eng, nxt, phase="head")
                        box["ev"] = torch.cuda.Event()
                        box["ev"].record(torch.cuda.current_stream())
                        keep.append(box["ev"])

                    def emit(nxt=nxt, box=box):
                        if "ev" in box:
                            torch.cuda.current_stream().wait_event(box["ev"])
                        else:       # (the engine had no side work for this batch: the head has not run yet)
                            synth.hash_tokens(eng, nxt, phase="head")
                        synth.hash_tokens(eng, nxt, phase="emit")

                    eng.train_step(tb.batch, beside_tower=head, before_join=emit)
            return
        main = torch.cuda.current_stream()
        keep = self._events = []
        ev_ids = []
        if not ids_input:
            s_h = eng._side(1)
            s_h.wait_stream(main)
            with torch.cuda.stream(s_h):
                for tb in tbs:
                    synth.hash_tokens(eng, tb)
                    ev = torch.cuda.Event()
                    ev.record(s_h)
                    keep.append(ev)
                    ev_ids.append(ev)
        for t, tb in enumerate(tbs):
            if ev_ids:
                main.wait_event(ev_ids[t])
                if t + 1 < len(tbs):
                    # step t also buckets batch t+1 (behind its dense tail): the graph needs the edge hash(t+1) -> that node
                    main.wait_event(ev_ids[t + 1])
            nxt =(tbs[t + 1].batch, (t + 1) & 1) if t + 1 < len(tbs) else None
            eng.train_step(tb.batch, pset=(t & 1) if t >= 1 else None, lookahead=nxt)
        if ev_ids:
            main.wait_stream(s_h)

    def _capture_pipelined(self, tbs, ids_input):
        eng = self.eng
        main = torch.cuda.current_stream()
        s_in, s_sp = eng._side(2), eng._side(0)
        s_in.wait_stream(main)              # fork: everything launched before this graph is complete
        s_sp.wait_stream(main)
        tail_after = os.environ.get("WD_PIPE_TAIL", "beside") == "after"
        bucket_early = os.environ.get("WD_PIPE_BUCKET", "early") == "early"
        ev_tower_prev = None
        pending = []
        keep = self._events = []            # every event lives as long as the graph (none is destroyed during the capture)

        def event(stream):
            ev = torch.cuda.Event()
            ev.record(stream)
            keep.append(ev)
            return ev

        # ---- hash branch: tokens -> ids of EVERY step of the graph, back to back at its head (WD_PIPE_HASH=ahead, default).
        # They depend on nothing; hashed one step ahead on the input branch (WD_PIPE_HASH=step) the hash of step t+1 lands in
        # the gather phase of tower(t) -- both latency-bound -- and takes 12-17 us there instead of 4.
        hash_ahead = os.environ.get("WD_PIPE_HASH", "ahead") == "ahead" and not ids_input
        ev_hash = []
        if hash_ahead:
            s_h = eng._side(1)
            s_h.wait_stream(main)
            with torch.cuda.stream(s_h):
                for tb in tbs:
                    synth.hash_tokens(eng, tb)
                    ev_hash.append(event(s_h))
        for t, tb in enumerate(tbs):
            bt = tb.batch
            eng._check_batch(bt)
            # ---- input branch: tokens -> ids (depends on nothing) ----
            if hash_ahead:
                s_in.wait_event(ev_hash[t])
                ev_ids = ev_hash[t]
            else:
                with torch.cuda.stream(s_in):
                    if not ids_input:
                        synth.hash_tokens(eng, tb)
                    ev_ids = event(s_in)
            # ---- bucketing: ids -> row-range buckets (scratch set t % 2).  WD_PIPE_BUCKET=early: on the input branch, released
            # by the END of tower(t-1) -- it then runs beside the products / update of step t-1 instead of beside tower(t),
            # whose one-workgroup-per-CU grid it slows down (99 us in the step against 90 alone)
            pset = t & 1
            if bucket_early:
                if t >= 1:
                    s_in.wait_event(ev_tower_prev)
                with torch.cuda.stream(s_in):
                    eng._sparse_bucketize(bt, s_in.cuda_stream, pset)
                    ev_buck = event(s_in)
            else:
                s_sp.wait_event(ev_ids)
                with torch.cuda.stream(s_sp):
                    eng._sparse_bucketize(bt, s_sp.cuda_stream, pset)
                ev_buck = None
            # ---- dense chain: tower(t) -> weight gradients || row update -> finalize + Adagrad + fold for t+1 ---------------
            main.wait_event(ev_ids)
            while pending:
                main.wait_event(pending.pop())          # WD_PIPE_TAIL=beside: the update of step t-1 is joined here
            eng.forward(bt, need_loss=True)             # folded weights are in place (eng._folded): the tower launch only
            ev_tower = event(main)
            ev_tower_prev = ev_tower
            # ---- weight-gradient GEMMs, then (captured AFTER them: ready nodes are launched in capture order, and the GEMMs'
            # 512 workgroups have to be resident before the update's 3200 flood the CUs -- launched together the GEMMs take 72 us
            # instead of 38) the row update on the sparse branch: Adagrad (embedding rows) + Ftrl (wide rows, bias).
            # The update is joined before tower(t+1) (default) or, WD_PIPE_TAIL=after, between the products and the tail.
            def update_then_join(ev_tower=ev_tower, ev_buck=ev_buck, pset=pset, bt=bt):
                s_sp.wait_event(ev_tower)
                if ev_buck is not None:
                    s_sp.wait_event(ev_buck)
                with torch.cuda.stream(s_sp):
                    eng._sparse_backward(bt, s_sp.cuda_stream, bucketized=True, pset=pset)
                    ev_upd = event(s_sp)
                if tail_after:
                    main.wait_event(ev_upd)
                else:
                    pending.append(ev_upd)

            eng._dense_backward(bt, main.cuda_stream, after_products=update_then_join)
        main.wait_stream(s_in)
        main.wait_stream(s_sp)
        if hash_ahead:
            main.wait_stream(s_h)

    def _capture_prefetched(self, tbs, ids_input):
        """The pipelined step with the input layer taken out of the tower launch (engine.prefetch): everything that needs only
        the ids of batch t+1 runs beside tower(t).  The sparse branch carries, in stream order,

            hash branch    tokens -> ids of every step of the graph, ahead
            sparse branch  update(t-1) | bucket(t+1) -> sort(t+1) -> prefetch(t+1) | update(t) + patch of x(t+1)
            main           tower(t) -> products(t) -> tail(t)                          tower(t) waits for update(t-1)

        prefetch(t+1) gathers the rows + wide weights of batch t+1 into the next activation buffer (wd_prefetch_onehot) from
        the tables as update(t-1) left them; update(t) -- which needs the sorted pairs of batches t and t+1 -- then stores the
        rows it rewrites into that buffer again (wd_apply_next_t), so tower(t+1), which reads x from HBM, sees exactly the
        tables after update(t).  No event sits between update(t-1) and tower(t) except the join itself; three scratch sets /
        activation buffers keep every writer behind the last reader.
        (A third graph branch for bucket + sort -- for skewed ids, whose sort is the long pole of the sparse branch -- was tried in
        round 3 and removed in round 4: ROCm 7.2's graph executor does not keep a captured stream on a hardware queue of its own,
        profiles/README.md.)  Every step's results are bit-identical to the eager launches
        (gather -> tower -> ...) in every layout, tests/test_gpu_prefetch.py."""
        eng = self.eng
        main = torch.cuda.current_stream()
        s_sp, s_h = eng._side(0), eng._side(1)
        s_sp.wait_stream(main)
        s_h.wait_stream(main)
        keep = self._events = []
        n = len(tbs)
        seq = list(tbs) + ([self.lookahead] if self.lookahead is not None else [])    # batches whose input work this graph does
        first = 1 if self.primed else 0                                               # ... from this one on
        set0, act0 = self.phase
        na, ns = eng.n_act, len(eng._bucket_sets)
        sset = lambda t: (set0 + t) % ns
        sact = lambda t: (act0 + t) % na

        def event(stream):
            ev = torch.cuda.Event()
            ev.record(stream)
            keep.append(ev)
            return ev

        # ---- ids branch: tokens -> ids of every batch, back to back at the head of the graph; then, batch by batch,
        # ids -> sorted (row, bag) pairs + the shared-row list of the batch before (sort_work below)
        ev_ids, ev_sort, ev_tail, ev_twr = {}, {}, {}, {}
        if not ids_input:
            with torch.cuda.stream(s_h):
                for t in range(first, len(seq)):
                    synth.hash_tokens(eng, seq[t])
                    ev_ids[t] = event(s_h)

        def sort_work(t):
            """bucket(t) + sort(t) -> scratch set t % 3 (+ which rows batch t-1 shares with batch t), on the ids branch.  The
            set's last reader, update(t-3), completed before tower(t-2) started; released by the end of tail(t-2), the work runs
            beside tower(t-1) -- and beside the gather of the same batch instead of in front of it (with skewed ids the sort of
            the head rows' buckets takes 80 us: in stream order in front of the gather it delayed update(t-1) by 30)."""
            bt = seq[t].batch
            eng._check_batch(bt)
            if t in ev_ids:
                s_sp.wait_event(ev_ids[t])
            with torch.cuda.stream(s_sp):
                eng._sparse_bucketize(bt, s_sp.cuda_stream, sset(t), prev=sset(t - 1) if t >= 1 else None)
                ev_sort[t] = event(s_sp)

        def gather_work(t):
            """prefetch(t) -> activation buffer t % 3 + wide weight list, on the sparse branch behind update(t-2) (stream order):
            the buffer's last readers -- tower and products of step t-3 -- completed before tower(t-2), which update(t-2)
            waited for."""
            bt = seq[t].batch
            if t in ev_ids:
                s_sp.wait_event(ev_ids[t])
            with torch.cuda.stream(s_sp):
                eng._prefetch_input(bt, s_sp.cuda_stream, sact(t))

        # diagnostics (timing only, results are garbage): WD_DIAG_SKIP=sort,gather,update leaves those launches out of a REPLAYED
        # step -- after the first batch of the graph, so that every buffer holds something -- to price what each costs the tower
        skip = set(filter(None, os.environ.get("WD_DIAG_SKIP", "").split(",")))
        if skip:
            sort0, gather0 = sort_work, gather_work
            sort_work = lambda t: sort0(t) if (t <= 1 or "sort" not in skip) else None
            gather_work = lambda t: gather0(t) if (t <= 1 or "gather" not in skip) else None
        # order of the two side jobs beside tower(t).  gather (default, round 5): the gather of batch t+1 first -- it meets the tower's
        # x tile (both read HBM: the x tile 17 k -> 29 k cycles) and leaves the head and B2 alone (21 k -> 13 k, 12 k -> 10 k): tower
        # -1 %, step -1.2 %, and the gather itself 12.8 -> 11.5 us in the step (roofline 0.28 -> 0.31).  WD_PIPE_SIDE=sort: rounds
        # 3-4, sort first (the gather then lands on the narrow stages and the head).  profiles/r5_tower_stage_cycles_in_step.txt
        # (bucketing first, then the gather, then the sort -- the gather under F0 instead of the x tile -- gives the x tile its 17-24 k
        # cycles back and takes 52-59 k for F0: the tower's 188-190 k cycles do not move, the step is 2 % slower)
        gather_first = os.environ.get("WD_PIPE_SIDE", "gather") == "gather"
        ev_upd = None
        if not self.primed:
            sort_work(0)
            gather_work(0)
            ev_upd = event(s_sp)        # x(0) in place
        for t, tb in enumerate(tbs):
            bt = tb.batch
            if ev_upd is not None:
                main.wait_event(ev_upd)                 # update(t-1), with its patch of this step's x
            if t + 1 < len(seq):
                if gather_first:
                    gather_work(t + 1)
                    sort_work(t + 1)
                else:
                    sort_work(t + 1)
                    gather_work(t + 1)
            eng._apar, eng._prefetched = sact(t), True
            eng.forward(bt, need_loss=True)             # the tower launch: x from HBM, wide logit from the weight list
            ev_tower = ev_twr[t] = event(main)
            hold = {}

            def update_then_join(t=t, bt=bt, ev_tower=ev_tower, hold=hold):
                s_sp.wait_event(ev_tower)
                nxt = (sset(t + 1), sact(t + 1)) if t + 1 < len(seq) else None
                with torch.cuda.stream(s_sp):
                    eng._sparse_backward(bt, s_sp.cuda_stream, bucketized=True, pset=sset(t), patch=nxt)
                    hold["upd"] = event(s_sp)

            eng._dense_backward(bt, main.cuda_stream, after_products=update_then_join)
            ev_tail[t] = event(main)
            ev_upd = hold["upd"]
        main.wait_stream(s_sp)
        main.wait_stream(s_h)

    @property
    def next_phase(self):
        """(bucket set, activation buffer) the `lookahead` batch is left in: the phase of the graph that continues the chain."""
        return ((self.phase[0] + self.n) % len(self.eng._bucket_sets), (self.phase[1] + self.n) % self.eng.n_act)

    def _token(self, tb, phase):
        return (id(tb), phase[0], phase[1], self.eng.global_step)

    def prime(self):
        """The input work of this graph's first batch as eager launches on the current stream (what the previous graph of a
        chain does through its `lookahead`): ids, sorted buckets (set phase[0]), x and the wide weight list (buffer phase[1])."""
        eng, tb = self.eng, self.first
        st = torch.cuda.current_stream().cuda_stream
        bt = tb.batch if self.ids_input else synth.hash_tokens(eng, tb)
        eng._check_batch(bt)
        eng._sparse_bucketize(bt, st, self.phase[0], prev=None)
        eng._prefetch_input(bt, st, self.phase[1])
        eng._primed = self._token(tb, self.phase)

    def replay(self):
        eng = self.eng
        if self.primed and getattr(eng, "_primed", None) != self._token(self.first, self.phase):
            self.prime()
        self.graph.replay()
        eng.global_step += self._bump
        eng._primed = self._token(self.lookahead, self.next_phase) if self.lookahead is not None else None
        return eng.loss


def warm(eng, token_batches, ids_input=False, steps=2):
    """Eager steps on a side stream (what torch needs before a capture); returns that stream."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(steps):
            step_eager(eng, token_batches[i % len(token_batches)], ids_input)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return side


class StepRunner:
    """What `bench.py` times: `steps` consecutive train steps as replays of multi-step hipGraphs over windows of the resident
    batch pool, plus one-step graphs for what does not fill a multi-step graph (warm-up steps, the remainder of an odd count).

    One hipGraph holds `spg` consecutive steps (each the full step on its own resident batch): the 10-30 us between two graph
    launches is paid once per `spg` steps.  spg = the largest divisor of `steps` up to `steps_per_graph` -- but at least two
    graphs for the timed region (the second is launched while the first runs; the driver's 20 timed steps are two replays of
    10).  On the prefetched single-GPU capture consecutive graphs are CHAINED like the steps inside one (StepGraph lookahead /
    phase / primed): a graph also does the input work (hash, buckets, sort, gather) of the next graph's first batch beside its
    last tower; six graphs close the cycle of (bucket set, activation buffer) phases for any spg (6 spg = 0 mod 2 and mod 3).

    `graph_cls` is StepGraph or dist.ShardedStepGraph (collectives captured).  Construction only CAPTURES; `warm_up()` replays
    every graph (untimed) -- with captured collectives the caller agrees across the ranks between the two that every capture
    succeeded, so that no rank replays collectives its peers never issue.  tests/test_gpu_fullsize.py builds the same object."""

    def __init__(self, eng, dev_batches, steps, steps_per_graph=32, ids_input=False, graph_cls=None, n_singles=None, stream=None):
        graph_cls = graph_cls or StepGraph
        self.eng, self.ids_input = eng, ids_input
        self.side = stream or warm(eng, dev_batches, ids_input)
        nb = len(dev_batches)
        cap = max(1, min(steps_per_graph, nb))
        spg = max(d for d in range(1, cap + 1) if steps % d == 0)
        if spg == steps and steps >= 16:
            spg = max(d for d in range(1, steps // 2 + 1) if steps % d == 0 and d <= cap)
        self.spg = spg
        self.chain = (spg > 1 and graph_cls is StepGraph and bool(getattr(eng, "prefetch", False))
                      and os.environ.get("WD_GRAPH_CHAIN", "1") != "0" and all(pipelined_ok(eng, tb) for tb in dev_batches))
        starts, j = [], 0
        while spg > 1 and (len(starts) < 6 if self.chain else (j not in starts and len(starts) < 8)):
            starts.append(j)
            j = (j + spg) % nb
        self.starts = starts
        if self.chain:
            self.multis, phase = [], (0, 0)
            for k, j0 in enumerate(starts):
                g = StepGraph(eng, [dev_batches[(j0 + i) % nb] for i in range(spg)], ids_input, stream=self.side,
                              lookahead=dev_batches[starts[(k + 1) % len(starts)]], phase=phase, primed=True)
                self.multis.append(g)
                phase = g.next_phase
            assert phase == (0, 0) and all(g.chained for g in self.multis)
        else:
            self.multis = [graph_cls(eng, [dev_batches[(j0 + i) % nb] for i in range(spg)], ids_input, stream=self.side)
                           for j0 in starts]
        self.singles = [graph_cls(eng, [tb], ids_input, stream=self.side) for tb in dev_batches[:n_singles or nb]]
        # multi-step graphs walk the pool forwards from batch 0, one-step graphs backwards from its end: a short run (the
        # driver's 20 steps after 5 warm-up steps) does not time batches whose rows the warm-up has just pulled into the
        # Infinity Cache
        self.cursor = {"m": 0, "s": 0}

    @property
    def pipelined(self):
        g = (self.multis or self.singles)[0]
        return bool(getattr(g, "pipelined", True))

    def warm_up(self, rounds=3):
        """clocks, caches and the graph executor's first-replay work out of the way: every graph is replayed (untimed; the
        first replay of a graph also pays its one-time upload)"""
        for rep in range(rounds):
            for g in self.multis + (self.singles[:4] if rep == 0 else []):
                g.replay()
        torch.cuda.synchronize()

    def run(self, n):
        spg = self.spg
        for _ in range(n // spg if spg > 1 else 0):
            self.multis[self.cursor["m"] % len(self.multis)].replay()
            self.cursor["m"] += 1
        for _ in range(n % spg if spg > 1 else n):
            self.singles[(len(self.singles) - 1 - self.cursor["s"]) % len(self.singles)].replay()
            self.cursor["s"] += 1
        if self.chain and n % spg:
            # the one-step graphs leave no input work behind: hand the chain what its next graph expects (the multi-step
            # graphs do this for each other; inside a timed region it happens exactly as often as outside)
            self.multis[self.cursor["m"] % len(self.multis)].prime()

    def next_batches(self, dev_batches):
        """the resident batches the next multi-step replay will train on, in order (for the tests' eager twin)"""
        j0 = self.starts[self.cursor["m"] % len(self.multis)]
        return [dev_batches[(j0 + i) % len(dev_batches)] for i in range(self.spg)]
