"""Command lines of the drop-in path: what `python train.py`, `python eval.py`, `python pred.py` run.

The three entry points keep the reference's flag names, defaults (conf/train.yaml), run modes and printed result lines
(python/train.py:27-62, 65-165, 188-229; python/eval.py:24-93; python/pred.py:24-77) so that scripts and log scrapers
written for the reference keep working.  Here the flags are ONE table (flag, type, conf key, help, which commands take it),
the three schedules are generators of (action, file) steps, and a single runner executes them on the Estimator-shaped
object of wide_deep_amd.estimator -- the TF session / tf.app machinery of the reference has no counterpart.

Multi-GPU training: one process per GPU,
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py ...
(replaces the parameter-server cluster of conf/train.yaml `distribution`; every rank reads lines i % N == rank).
The image-tower flags are accepted for CLI compatibility and must stay empty (CNN towers are out of scope)."""
import argparse
import os
import shutil
import sys
import time

from .build_estimator import build_custom_estimator
from .dataset import input_fn, list_files
from .read_conf import Config

T, E, P = "train", "eval", "pred"
# flag, type, key in conf/train.yaml (None: no default there), help, commands that accept it
FLAGS = [
    ("model_dir", str, "model_dir", "Base directory for the model (checkpoints live in <model_dir>/<model_type>).", (T, E, P)),
    ("model_type", str, "model_type", "Valid model types: {'wide', 'deep', 'wide_deep'}.", (T, E, P)),
    ("train_epochs", int, "train_epochs", "Number of training epochs.", (T,)),
    ("epochs_per_eval", int, "epochs_per_eval", "The number of training epochs to run between evaluations.", (T,)),
    ("batch_size", int, "batch_size", "Number of examples per batch.", (T, E, P)),
    ("train_data", str, "train_data", "Path to the train data.", (T,)),
    ("eval_data", str, "eval_data", "Path to the validation data.", (T,)),
    ("test_data", str, "test_data", "Path to the test data.", (T, E)),
    ("image_train_data", str, "image_train_data", "(image tower: out of scope, keep empty)", (T,)),
    ("image_eval_data", str, "image_eval_data", "(image tower: out of scope, keep empty)", (T,)),
    ("image_test_data", str, "image_test_data", "(image tower: out of scope, keep empty)", (T, E)),
    ("keep_train", int, "keep_train", "Whether to keep training on previous trained model.", (T,)),
    ("data_dir", str, None, "Prediction data file or dir.", (P,)),
    ("image_data_dir", str, None, "(image tower: out of scope, keep empty)", (P,)),
    ("checkpoint_path", str, "checkpoint_path", "A specific checkpoint; if empty, the latest one in model_dir is used.", (E, P)),
]
TITLES = {T: "Train Wide and Deep Model.", E: "Evaluate Wide and Deep Model.", P: "Wide and Deep Model Prediction"}


def build_parser(command, cfg):
    p = argparse.ArgumentParser(description=TITLES[command])
    for flag, typ, key, text, commands in FLAGS:
        if command in commands:
            p.add_argument("--" + flag, type=typ, default=cfg.get(key) if key else None, help=text)
    return p


def minutes_since(t0):
    return round((time.time() - t0) / 60, 2)


def _print_metrics(results, out):
    for key in sorted(results):
        print("{}: {}".format(key, results[key]), file=out)


def _throughput(model, tag, out):
    lt = getattr(model, "last_train", None)
    if lt:
        print("INFO: %s: %d steps, %d examples, %.1f examples/sec, last batch loss %.6f" % (
            tag, lt["steps"], lt["examples"], lt["examples"] / max(lt["seconds"], 1e-9), lt["loss"]), file=out)


# ---- the three schedules of python/train.py as step generators: (action, data, label for the log) ---------------------
def schedule_train_and_eval(F):
    """every epoch: each train file, the validation set after each; the test set every `epochs_per_eval` epochs (train.py:65-93).
    The reference hands mode 'pred' to that last evaluate (train.py:98), which carries no labels; the test set is evaluated
    with its labels here."""
    for n in range(F.train_epochs):
        yield ("banner", None, " START EPOCH {} ".format(n + 1))
        for f in list_files(F.train_data):
            yield (T, f, "<EPOCH {}>: training {}".format(n + 1, f))
            yield (E, F.eval_data, "<EPOCH {}>: evaluation {}".format(n + 1, F.eval_data))
        if (n + 1) % F.epochs_per_eval == 0:
            yield (E, F.test_data, "<EPOCH {}>: testing {}".format(n + 1, F.test_data))


def schedule_dynamic(F):
    """files in name order: train on file i for `train_epochs`, evaluating on file i+1 after every epoch (train.py:96-148)"""
    files = sorted(list_files(F.train_data))
    assert len(files) > 1, "Dynamic train mode need more than 1 data file"
    for cur, nxt in zip(files[:-1], files[1:]):
        yield ("banner", None, " START TRAINING DATA: {} ".format(cur))
        for n in range(F.train_epochs):
            yield (T, cur, "TRAIN DATA <{}> <EPOCH {}>".format(cur, n + 1))
            yield (E, nxt, "EVALUATE TEST DATA <{}> <EPOCH {}>".format(nxt, n + 1))


def schedule_train_only(F):
    """distributed runs do not evaluate (train.py:151-165, 213-214)"""
    for n in range(F.train_epochs):
        yield ("banner", None, " START EPOCH {} ".format(n + 1))
        for f in list_files(F.train_data):
            yield (T, f, "<EPOCH {}>: training {}".format(n + 1, f))


def run_schedule(model, F, steps, out=None):
    out = out or sys.stdout
    image = {T: getattr(F, "image_train_data", None), E: getattr(F, "image_eval_data", None)}
    for action, data, label in steps:
        if action == "banner":
            print("=" * 30 + label + "=" * 30 + "\n", file=out)
            continue
        t0 = time.time()
        print("START " + label, file=out)
        if action == T:
            model.train(input_fn=lambda: input_fn(data, image[T], "train", F.batch_size))
            print("FINISH {}, take {} mins".format(label, minutes_since(t0)), file=out)
            _throughput(model, data, out)
        else:
            results = model.evaluate(input_fn=lambda: input_fn(data, image[E], "eval", F.batch_size))
            print("FINISH {}, take {} mins".format(label, minutes_since(t0)), file=out)
            print("-" * 80, file=out)
            _print_metrics(results, out)
        print("-" * 80, file=out)


def _header(F, conf, out):
    print("\nModel Type: {}".format(F.model_type), file=out)
    model_dir = os.path.join(F.model_dir, F.model_type)
    print("\nModel Directory: {}".format(model_dir), file=out)
    return model_dir


def train_main(argv=None, out=None):
    out = out or sys.stdout
    conf = Config()
    F, _ = build_parser(T, conf.train).parse_known_args(argv)
    model_dir = _header(F, conf, out)
    for title, section in (("Train", conf.train), ("Model", conf.model)):
        print("\nUsing {} Config:".format(title), file=out)
        for k, v in section.items():
            print("{}: {}".format(k, v), file=out)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    if not F.keep_train and int(os.environ.get("RANK", "0")) == 0:
        shutil.rmtree(model_dir, ignore_errors=True)
        print("Remove model directory: {}".format(model_dir), file=out)
    if world > 1:
        dist.barrier()      # nobody looks for a checkpoint before rank 0 has removed the old directory
    model = build_custom_estimator(model_dir, F.model_type, conf=conf, max_batch=F.batch_size)
    if world > 1 or conf.distribution.get("is_distribution"):
        steps = schedule_train_only(F)
    elif conf.train["dynamic_train"]:
        print("Using dynamic train mode.", file=out)
        steps = schedule_dynamic(F)
    else:
        steps = schedule_train_and_eval(F)
    run_schedule(model, F, steps, out)
    return model


def eval_main(argv=None, out=None):
    out = out or sys.stdout
    conf = Config()
    F, _ = build_parser(E, conf.train).parse_known_args(argv)
    model_dir = _header(F, conf, out)
    model = build_custom_estimator(model_dir, F.model_type, max_batch=F.batch_size)
    print("INFO: " + "=" * 30 + " START TESTING " + "=" * 30, file=out)
    t0 = time.time()
    results = model.evaluate(input_fn=lambda: input_fn(F.test_data, F.image_test_data or None, "eval", F.batch_size),
                             checkpoint_path=F.checkpoint_path or None)
    print("INFO: " + "=" * 30 + " FINISH TESTING, TAKE {} mins ".format(minutes_since(t0)) + "=" * 30, file=out)
    print("-" * 80, file=out)
    _print_metrics(results, out)
    return results


def pred_main(argv=None, out=None):
    out = out or sys.stdout
    conf = Config()
    F, _ = build_parser(P, conf.train).parse_known_args(argv)
    if F.data_dir is None:
        raise ValueError("Must specify prediction data_file by --data_dir")
    model_dir = _header(F, conf, out)
    model = build_custom_estimator(model_dir, F.model_type, max_batch=F.batch_size)
    print("INFO: " + "=" * 30 + " START PREDICTION " + "=" * 30, file=out)
    t0, n = time.time(), 0
    rows = model.predict(input_fn=lambda: input_fn(F.data_dir, F.image_data_dir, "pred", F.batch_size),
                         checkpoint_path=F.checkpoint_path or None)
    for row in rows:                       # {logits, logistic, probabilities, class_ids, classes}
        winner = int(row["class_ids"][0])
        print('\nPrediction is "{}" ({:.1f}%)'.format(winner, 100 * float(row["probabilities"][winner])), file=out)   # pred.py:71-74
        n += 1
    print("INFO: " + "=" * 30 + " FINISH PREDICTION of {} rows, TAKE {} mins ".format(n, minutes_since(t0)) + "=" * 30, file=out)
    return n
