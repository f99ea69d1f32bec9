"""Estimator-shaped front of the engine: what python/train.py drives (python/lib/joint.py:272-432).

`WideAndDeepClassifier(spec, model_dir)` offers the three calls the reference's drivers use,
    .train(input_fn, hooks=None, steps=None, max_steps=None, saving_listeners=None)        train.py:72-77,128-133
    .evaluate(input_fn, steps=None, hooks=None, checkpoint_path=None, name=None) -> dict   train.py:81-87,138-143
    .predict(input_fn, predict_keys=None, hooks=None, checkpoint_path=None) -> iterator    pred.py:65-74
with the Estimator's life cycle: every call restores the latest checkpoint of `model_dir` (if any), `train` runs
until the one-pass input is exhausted and writes a checkpoint at the end (timed saves every
`save_checkpoints_secs`, `keep_checkpoint_max` newest kept).  Checkpoints are `model.ckpt-<global_step>.pt` files
holding a {TF variable name -> tensor} dict in the reference's naming, so weights can be moved
to / from a TF checkpoint of the reference by name.

`input_fn` is a zero-argument callable returning an iterator of dataset.RawBatch (as `lambda: input_fn(...)` in
train.py) -- or of ready DeviceBatch objects (synthetic benches).
"""
import glob
import os
import re
import time

import numpy as np
import torch

from .engine import DeviceBatch, WideDeepEngine
from .features import Featurizer


class _GraphStep(object):
    """`Featurizer.run` + `engine.train_step` of ONE batch size captured into a hipGraph over fixed-capacity buffers
    (features.FixedStage): a train step of the loop is then one staged host-to-device copy + one graph launch instead of the
    featurizer's and the step's ~80 eager launches (python/train.py:65-165 at the reference's batch sizes -- 64 shipped,
    conf/train.yaml:47; BASELINE configs[0] 512 -- is launch-bound on this GPU, not kernel-bound).  The id count and the token
    count stay on the device; `check()` reads the capacity flags every few steps."""

    def __init__(self, est, raws):
        import os
        from . import hipgraph
        eng, fz = est._engine, est._featurizer
        B = raws[0].B
        tok = max(len(r.tok_offs) - 2 for r in raws)
        nby = max(len(r.tok_bytes) for r in raws)
        slack = float(os.environ.get("WD_TRAIN_GRAPH_SLACK", "1.5"))
        self.eng, self.fz, self.B = eng, fz, B
        self.pdb = fz.resident_fixed(B, int(tok * slack) + 256, int(nby * slack) + 4096, labels=raws[0].labels is not None,
                                     weights=raws[0].weights is not None and eng.spec.use_weight_column,
                                     nnz_hint=est._nnz_seen * 1.25 + 1024 if est._nnz_seen else None)
        self.stage = self.pdb.stg
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.stage.fill(raws[-1])
            fz.run(self.pdb)              # (first launches of the dynamic-count kernels outside the capture; changes no state)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = hipgraph.new_graph()
        gs = eng.global_step
        with torch.cuda.graph(self.graph, stream=side):
            fz.run(self.pdb)
            eng.train_step(self.pdb.batch)
        eng.global_step = gs
        self._bump = 3 if eng.spec.model_type == "wide_deep" else 2
        self.stream = side

    def fits(self, raw):
        return self.stage.fits(raw)

    def step(self, raw):
        self.stage.fill(raw)
        self.graph.replay()
        self.eng.global_step += self._bump
        return self.eng.loss

    def check(self):
        """the id array's capacity flag (wd_feat_offsets): a batch with more ids than the engine holds dropped some"""
        if int(self.pdb.flags[0].item()):
            raise ValueError("a batch produced more ids than the engine's capacity (max_nnz=%d)" % self.eng.max_nnz)


class WideAndDeepClassifier(object):
    def __init__(self, spec, model_dir=None, runconfig=None, max_batch=None, max_nnz=None, cross_padding="tf_dense",
                 seed=None, engine=None):
        self.spec = spec
        self.model_dir = model_dir
        self.runconfig = dict(runconfig or {})
        self._engine_kw = dict(max_batch=max_batch, max_nnz=max_nnz)
        self._seed = self.runconfig.get("tf_random_seed", 0) if seed is None else seed
        self._cross_padding = cross_padding
        self._engine = engine
        self._featurizer = None
        self._restored_from = None
        self._graph_steps = {}        # batch size -> _GraphStep (captured featurizer + train step)
        self._warm = {}               # batch size -> the first RawBatches of that size (eager steps; they size the fixed stage)
        self._nnz_seen = 0

    # ---- engine life cycle --------------------------------------------------------------------------
    def _ensure_engine(self, first_batch):
        if self._engine is None:
            B = self._engine_kw["max_batch"] or max(first_batch.B, 1)
            kw = {"max_batch": B, "seed": int(self._seed or 0)}
            if self._engine_kw["max_nnz"]:
                kw["max_nnz"] = self._engine_kw["max_nnz"]
            else:
                # multi-valued + crossed columns: generous capacity, the featurizer checks every batch against it
                kw["max_nnz"] = B * max(len(self.spec.slots), 1) * 16
            if self._world() > 1:
                # one process per GPU (python -m torch.distributed.run ... train.py): tables row-sharded, dense gradients
                # all-reduced -- N ranks train ONE model on the union of their batches (wide_deep_amd/dist.py)
                from .dist import ShardedWideDeepEngine
                # per-peer exchange segments sized from what a batch really holds (first batch + 50 % slack inside the
                # engine), not from the max_nnz buffer bound: the all-to-alls always move FULL segments.  A later, much
                # larger batch trips the overflow flag, which train / evaluate / predict check (check_overflow)
                nnz0 = int(getattr(first_batch, "nnz", 0) or 0)
                if nnz0 == 0 and hasattr(first_batch, "B"):
                    nnz0 = self._featurizer_nnz_estimate(first_batch)
                if nnz0:
                    kw["expected_nnz"] = min(int(nnz0 * 1.25) + 1024, kw["max_nnz"])
                self._engine = ShardedWideDeepEngine(self.spec, **kw)
            else:
                self._engine = WideDeepEngine(self.spec, **kw)
        if self._featurizer is None:
            self._featurizer = Featurizer(self._engine, self._cross_padding)
        return self._engine

    @property
    def engine(self):
        return self._engine

    def _featurizer_nnz_estimate(self, raw):
        """occurrences of a raw (host) batch before a featurizer exists: bags x a generous mean length (multi-valued features
        and their crosses; the reference data has 1.0-3.5 values per feature).  Too small -> check_overflow raises."""
        return int(getattr(raw, "B", 0)) * max(len(self.spec.slots), 1) * 4

    _CHECK_EVERY = 64     # steps between the periodic multi-rank checks (one device->host read / one broadcast each)

    def _check_overflow(self):
        """Row-sharded engine: a peer segment that overflowed dropped occurrences (include/wd_hip.h) -- never silently."""
        if self._engine is not None and hasattr(self._engine, "check_overflow"):
            self._engine.check_overflow()

    def _save_due(self, n, t_save, save_secs):
        """Timed checkpoints.  One process: the local clock.  Several ranks: saving is a collective, so rank 0's clock
        decides for everybody, every _CHECK_EVERY steps (a rank-local decision would desynchronise the collectives)."""
        if not save_secs:
            return False
        if self._world() == 1:
            return time.time() - t_save >= save_secs
        if n % self._CHECK_EVERY:
            return False
        import torch.distributed as td
        dev = self._engine.device if td.get_backend() == "nccl" else "cpu"
        flag = torch.tensor([1 if (self._rank() == 0 and time.time() - t_save >= save_secs) else 0], dtype=torch.int32, device=dev)
        td.broadcast(flag, src=0)
        return bool(int(flag.item()))

    _GRAPH_AFTER = 2      # eager steps of a batch size before its step is captured (lazy allocations, capacities from real batches)

    def _graph_step_for(self, raw):
        """The captured step for this batch, or None (eager): one GPU, parsed host batches through the device featurizer, constant
        learning rates (a captured step bakes them in), WD_TRAIN_GRAPH != 0, batches up to WD_TRAIN_GRAPH_MAX_BATCH (2048) examples --
        the small batches the reference ships are bound by the ~80 launches of a step, a batch of 8192 by the kernels and the
        parser, and its eager featurizer overlaps the previous step on a stream of its own (measured: 64: +39 %, 512: +22 %,
        8192: -8 %, profiles/r6_c1_train_loop.md); a batch size is captured once it has been seen _GRAPH_AFTER times (the last,
        shorter batch of a file stays eager unless it comes back every epoch)."""
        import os
        if (isinstance(raw, DeviceBatch) or self._engine is None or self._world() > 1 or self._engine.spec.lr_decay
                or self._featurizer is None or self._featurizer.mode != "device" or os.environ.get("WD_TRAIN_GRAPH", "1") == "0"
                or raw.B == 0 or raw.labels is None or raw.B > int(os.environ.get("WD_TRAIN_GRAPH_MAX_BATCH", "2048"))):
            return None
        g = self._graph_steps.get(raw.B)
        if g is None:
            w = self._warm.setdefault(raw.B, [])
            if len(w) < self._GRAPH_AFTER:
                w.append(raw)
                return None
            try:
                g = self._graph_steps[raw.B] = _GraphStep(self, w + [raw])
            except Exception as e:         # a capture the runtime refuses: stay eager for this size, say why once
                import warnings
                warnings.warn("train: the step of batch size %d is not captured into a hipGraph (%s: %s); eager launches"
                              % (raw.B, type(e).__name__, e))
                g = self._graph_steps[raw.B] = False
            self._warm.pop(raw.B, None)
        if not g or not g.fits(raw):
            return None                    # (more tokens than the fixed stage holds: this batch takes the eager path)
        return g

    def _device_batch(self, b):
        if isinstance(b, DeviceBatch):
            self._ensure_engine(b)
            return b
        self._ensure_engine(b)
        return self._featurizer.to_device(b)

    # ---- checkpoints --------------------------------------------------------------------------------
    def latest_checkpoint(self):
        if not self.model_dir or not os.path.isdir(self.model_dir):
            return None
        best, best_step = None, -1
        for p in glob.glob(os.path.join(self.model_dir, "model.ckpt-*.pt")):
            m = re.search(r"model\.ckpt-(\d+)\.pt$", p)
            if m and int(m.group(1)) > best_step:
                best, best_step = p, int(m.group(1))
        if best is None:
            # nothing of ours: a checkpoint the reference's tf.estimator left in model_dir (python/train.py:188-191: `keep_train`
            # resumes from it) -- TensorFlow's tensor-bundle container, read by wide_deep_amd/tf_checkpoint.py
            from .tf_checkpoint import latest_tf_checkpoint
            prefix = latest_tf_checkpoint(self.model_dir)
            if prefix:
                return prefix + ".index"
        return best

    @staticmethod
    def _world():
        import torch.distributed as td
        return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1

    @staticmethod
    def _rank():
        import torch.distributed as td
        return td.get_rank() if (td.is_available() and td.is_initialized()) else 0

    def _restore(self, checkpoint_path=None):
        path = checkpoint_path or self.latest_checkpoint()
        if path and path != self._restored_from:
            if path.endswith(".index"):
                from .tf_checkpoint import read_tf_checkpoint
                state = {k: torch.from_numpy(v) for k, v in read_tf_checkpoint(path[:-len(".index")]).items()}
                self._check_tf_state(state, path)
            else:
                state = torch.load(path, map_location="cpu")
            if self._world() > 1:
                self._engine.import_full_state(state)      # checkpoints hold FULL tables; a rank keeps its rows
                self._engine.global_step = int(state.get("global_step", 0))
            else:
                self._engine.import_state(state)
            self._restored_from = path
        return path

    def _check_tf_state(self, state, path):
        """A TensorFlow bundle is matched BY NAME: a variable this model expects that the file does not hold (a naming or
        partitioning difference) must not silently resume from fresh weights while global_step is restored -- raise, naming
        what is missing; shapes are checked before anything is copied; variables of the file nobody reads are listed."""
        if self._world() > 1:
            gp = self._engine.global_plan
            rows = {}
            for sl in gp.slots:
                rows["dnn/input_from_feature_columns/input_layer/%s/embedding_weights" % sl.deep_name] = int(sl.num_buckets)
                rows["linear/linear_model/%s/weights" % sl.name] = int(sl.num_buckets)
            want = {}
            for k, shape in self._engine.state_shapes().items():
                base = k
                for suf in ("/Adagrad", "/Ftrl_1", "/Ftrl", "/RMSProp_2", "/RMSProp_1", "/RMSProp", "/Adam_1", "/Adam"):
                    if base.endswith(suf):
                        base = base[: -len(suf)]
                        break
                if base in rows:
                    shape = (rows[base],) + shape[1:]
                want[k] = shape
        else:
            want = self._engine.state_shapes()
        skip = lambda k: k == "global_step" or k.endswith("/moving_mean") or k.endswith("/moving_variance")
        missing = sorted(k for k in want if k not in state and not skip(k))
        if missing:
            raise ValueError("%s does not hold %d of the variables this model restores by name (first: %s); variables of the "
                             "file: %s ..." % (path, len(missing), ", ".join(missing[:4]), ", ".join(sorted(state)[:4])))
        bad = [(k, tuple(state[k].shape), want[k]) for k in want
               if k in state and not skip(k) and int(np.prod(state[k].shape)) != int(np.prod(want[k]))]
        if bad:
            raise ValueError("%s: shape mismatch for %s" % (path, "; ".join("%s is %s, the model has %s" % b for b in bad[:4])))
        unused = sorted(k for k in state if k not in want)
        if unused and self._rank() == 0:
            print("INFO: %d variables of %s are not part of this model and are ignored (first: %s)"
                  % (len(unused), path, ", ".join(unused[:4])))

    def save_checkpoint(self):
        if not self.model_dir:
            return None
        path = os.path.join(self.model_dir, "model.ckpt-%d.pt" % self._engine.global_step)
        if self._world() > 1:
            import torch.distributed as td
            state = self._engine.export_full_state()        # collective: every rank takes part, rank 0 writes
            if self._rank() == 0:
                os.makedirs(self.model_dir, exist_ok=True)
                torch.save(state, path)
            td.barrier()
            self._restored_from = path
            if self._rank() != 0:
                return path
        else:
            os.makedirs(self.model_dir, exist_ok=True)
            torch.save(self._engine.export_state(), path)
            self._restored_from = path
        keep = int(self.runconfig.get("keep_checkpoint_max") or 5)
        ckpts = sorted(glob.glob(os.path.join(self.model_dir, "model.ckpt-*.pt")),
                       key=lambda p: int(re.search(r"-(\d+)\.pt$", p).group(1)))
        for old in ckpts[:-keep]:
            os.remove(old)
        return path

    def export_tf_checkpoint(self, prefix=None):
        """The current state in TensorFlow's checkpoint container (`<prefix>.index` + `.data-00000-of-00001`, variable names of
        the reference), e.g. to hand weights trained here to the reference's eval / serving code."""
        from .tf_checkpoint import write_tf_checkpoint
        prefix = prefix or os.path.join(self.model_dir, "model.ckpt-%d" % self._engine.global_step)
        state = self._engine.export_full_state() if self._world() > 1 else self._engine.export_state()
        if self._rank() == 0:
            write_tf_checkpoint(prefix, {k: v.cpu().numpy() for k, v in state.items()})
        return prefix

    # ---- train / evaluate / predict -----------------------------------------------------------------
    def train(self, input_fn, hooks=None, steps=None, max_steps=None, saving_listeners=None):
        it = iter(input_fn())
        n, t_save, t0, seen = 0, time.time(), time.time(), 0
        # WD_TRAIN_TIMES=1: where the host thread of the loop spends its time (last_train["host_seconds"]: waiting for the parser,
        # staging a batch, launching its step) -- diagnostics of scripts/bench_c1.py
        prof = {"parse_wait": 0.0, "stage": 0.0, "launch": 0.0} if os.environ.get("WD_TRAIN_TIMES") == "1" else None
        if prof is not None:
            def timed_iter(src):
                while True:
                    t = time.perf_counter()
                    try:
                        x = next(src)
                    except StopIteration:
                        return
                    prof["parse_wait"] += time.perf_counter() - t
                    yield x
            it = timed_iter(it)
        save_secs = self.runconfig.get("save_checkpoints_secs")
        log_every = int(self.runconfig.get("log_step_count_steps") or 0)
        loss = None
        for raw in it:
            gstep = self._graph_step_for(raw) if n else None
            bt = None
            if gstep is None:
                bt = self._device_batch(raw)
            if n == 0:
                self._restore()
                t0 = time.time()       # examples/sec of the loop itself: checkpoint restore / save are reported apart
                if max_steps is not None and self._engine.global_step >= max_steps:
                    break
            if self._engine.spec.lr_decay:      # opt-in (train.yaml lr_decay: true): exponential decay over TF's global step
                eng = self._engine
                sp, gs = eng.spec, eng.global_step
                eng.set_learning_rates(dnn=sp.decayed_lr("dnn", gs, eng.lr0["dnn"]) if sp.has_deep else None,
                                       linear=sp.decayed_lr("linear", gs, eng.lr0["linear"]) if sp.has_wide else None)
            if gstep is not None and prof is not None:
                t1 = time.perf_counter()
                gstep.stage.fill(raw)
                t2 = time.perf_counter()
                gstep.graph.replay()
                self._engine.global_step += gstep._bump
                loss = self._engine.loss
                prof["stage"] += t2 - t1
                prof["launch"] += time.perf_counter() - t2
            elif gstep is not None:
                loss = gstep.step(raw)          # one staged copy + one hipGraph launch: featurizer + train step
                if n % self._CHECK_EVERY == 0:
                    gstep.check()
            else:
                loss = self._engine.train_step(bt)
                self._nnz_seen = max(self._nnz_seen, int(getattr(bt, "nnz", 0) or 0))
            n += 1
            seen += raw.B if bt is None else bt.B
            if log_every and n % log_every == 0 and self._rank() == 0:
                torch.cuda.synchronize()
                dt = time.time() - t0
                print("INFO: step %d (global_step %d): loss = %.6f, %.1f examples/sec" % (
                    n, self._engine.global_step, float(loss), seen / max(dt, 1e-9)))
            if n % self._CHECK_EVERY == 0:
                self._check_overflow()
            if self._save_due(n, t_save, save_secs):
                self._check_overflow()
                self.save_checkpoint()
                t_save = time.time()
            if steps is not None and n >= steps:
                break
            if max_steps is not None and self._engine.global_step >= max_steps:
                break
        if n:
            torch.cuda.synchronize()
            self._check_overflow()
            for g in self._graph_steps.values():
                if g:
                    g.check()
            self.last_train = {"steps": n, "examples": seen, "seconds": time.time() - t0,
                               "loss": float(loss) if loss is not None else None,
                               "graph_batch_sizes": sorted(b for b, g in self._graph_steps.items() if g)}
            if prof is not None:
                self.last_train["host_seconds"] = {k: round(v, 4) for k, v in prof.items()}
            self.save_checkpoint()
        return self

    def _forward_all(self, input_fn, steps, checkpoint_path, need_labels):
        probs, logits, labels, weights = [], [], [], []
        n = 0
        for raw in iter(input_fn()):
            bt = self._device_batch(raw)
            if n == 0:
                self._restore(checkpoint_path)
            self._engine.forward(bt, need_loss=False)
            B = bt.B
            probs.append(self._engine.prob[:B].cpu())
            logits.append(self._engine.logit[:B].cpu())
            if need_labels:
                if bt.labels is None:
                    raise ValueError("evaluate needs labels (input_fn mode 'eval')")
                labels.append(bt.labels[:B].cpu())
                weights.append(bt.weights[:B].cpu() if bt.weights is not None else torch.ones(B))
            n += 1
            if steps is not None and n >= steps:
                break
        if n:
            self._check_overflow()
        cat = lambda xs: torch.cat(xs) if xs else torch.zeros(0)
        return cat(probs), cat(logits), cat(labels), cat(weights)

    def evaluate(self, input_fn, steps=None, hooks=None, checkpoint_path=None, name=None):
        """Metrics of the canned binary head: accuracy, accuracy_baseline, auc, auc_precision_recall,
        average_loss, label/mean, loss, precision, prediction/mean, recall, global_step."""
        p, x, y, w = self._forward_all(input_fn, steps, checkpoint_path, True)
        return binary_head_metrics(p.double().numpy(), x.double().numpy(), y.double().numpy(), w.double().numpy(),
                                   self._engine.global_step if self._engine else 0, self._last_batch_size(p))

    def _last_batch_size(self, p):
        return self._engine.max_batch if self._engine else max(len(p), 1)

    def predict(self, input_fn, predict_keys=None, hooks=None, checkpoint_path=None):
        p, x, _, _ = self._forward_all(input_fn, None, checkpoint_path, False)
        for i in range(len(p)):
            pi, xi = float(p[i]), float(x[i])
            out = {"logits": np.asarray([xi], np.float32), "logistic": np.asarray([pi], np.float32),
                   "probabilities": np.asarray([1.0 - pi, pi], np.float32),
                   "class_ids": np.asarray([int(pi > 0.5)], np.int64),
                   "classes": np.asarray([str(int(pi > 0.5)).encode()], dtype=object)}
            yield out if predict_keys is None else {k: out[k] for k in predict_keys}


def _auc(tp, fp, tn, fn, curve):
    """tf.metrics.auc(num_thresholds=200, summation_method='trapezoidal')."""
    eps = 1e-6
    if curve == "ROC":
        xs = fp / (fp + tn + eps)
        ys = (tp + eps) / (tp + fn + eps)
    else:
        xs = (tp + eps) / (tp + fn + eps)
        ys = (tp + eps) / (tp + fp + eps)
    return float(np.sum((xs[:-1] - xs[1:]) * (ys[:-1] + ys[1:]) / 2.0))


def binary_head_metrics(p, logits, y, w, global_step, batch_size):
    n = len(p)
    if n == 0:
        return {"global_step": global_step}
    wsum = w.sum()
    ce = np.maximum(logits, 0) - logits * y + np.log1p(np.exp(-np.abs(logits)))
    pred = (p > 0.5).astype(np.float64)
    tp = float((w * pred * y).sum())
    fp = float((w * pred * (1 - y)).sum())
    fn = float((w * (1 - pred) * y).sum())
    # 200 thresholds: [0-eps, 1/199, ..., 198/199, 1+eps]  (tf.metrics.auc)
    k = 200
    th = np.asarray([0.0 - 1e-7] + [(i + 1) * 1.0 / (k - 1) for i in range(k - 2)] + [1.0 + 1e-7])
    above = p[None, :] > th[:, None]
    tps = (above * (w * y)[None, :]).sum(1)
    fps = (above * (w * (1 - y))[None, :]).sum(1)
    fns = ((~above) * (w * y)[None, :]).sum(1)
    tns = ((~above) * (w * (1 - y))[None, :]).sum(1)
    label_mean = float((w * y).sum() / wsum)
    nb = int(np.ceil(n / float(batch_size)))
    return {
        "accuracy": float((w * (pred == y)).sum() / wsum),
        "accuracy_baseline": max(label_mean, 1.0 - label_mean),
        "auc": _auc(tps, fps, tns, fns, "ROC"),
        "auc_precision_recall": _auc(tps, fps, tns, fns, "PR"),
        "average_loss": float((w * ce).sum() / wsum),
        "label/mean": label_mean,
        "loss": float((w * ce).sum() / max(nb, 1)),      # mean over batches of the per-batch SUM loss
        "precision": tp / (tp + fp) if tp + fp > 0 else 0.0,
        "prediction/mean": float((w * p).sum() / wsum),
        "recall": tp / (tp + fn) if tp + fn > 0 else 0.0,
        "global_step": global_step,
    }
