#!/usr/bin/env python
"""`python train.py` -- the reference's training entry point (python/train.py:24-253) on the MI355X engine.

Same flags (defaults from conf/train.yaml), same run modes: `dynamic_train` (train.yaml default: train on file i for
`train_epochs`, evaluate on file i+1, train.py:96-148), `train_and_eval` (train.py:65-93) and plain `train` for the
distributed case.  Multi-GPU: launch one process per GPU with
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py ...
(replaces the TF parameter-server cluster of conf/train.yaml `distribution`; every rank reads lines i % N == rank).
The image-tower flags are accepted for CLI compatibility and must stay empty.
"""
import argparse
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from wide_deep_amd.build_estimator import build_custom_estimator  # noqa: E402
from wide_deep_amd.dataset import input_fn, list_files  # noqa: E402
from wide_deep_amd.read_conf import Config  # noqa: E402


def elapse_time(t0):
    return round((time.time() - t0) / 60, 2)   # minutes (python/lib/utils/util.py:32-33)


def build_parser(cfg):
    p = argparse.ArgumentParser(description="Train Wide and Deep Model.")
    p.add_argument("--model_dir", type=str, default=cfg["model_dir"], help="Base directory for the model.")
    p.add_argument("--model_type", type=str, default=cfg["model_type"], help="Valid model types: {'wide', 'deep', 'wide_deep'}.")
    p.add_argument("--train_epochs", type=int, default=cfg["train_epochs"], help="Number of training epochs.")
    p.add_argument("--epochs_per_eval", type=int, default=cfg["epochs_per_eval"],
                   help="The number of training epochs to run between evaluations.")
    p.add_argument("--batch_size", type=int, default=cfg["batch_size"], help="Number of examples per batch.")
    p.add_argument("--train_data", type=str, default=cfg["train_data"], help="Path to the train data.")
    p.add_argument("--eval_data", type=str, default=cfg["eval_data"], help="Path to the validation data.")
    p.add_argument("--test_data", type=str, default=cfg["test_data"], help="Path to the test data.")
    p.add_argument("--image_train_data", type=str, default=cfg.get("image_train_data"))
    p.add_argument("--image_eval_data", type=str, default=cfg.get("image_eval_data"))
    p.add_argument("--image_test_data", type=str, default=cfg.get("image_test_data"))
    p.add_argument("--keep_train", type=int, default=cfg["keep_train"],
                   help="Whether to keep training on previous trained model.")
    return p


def _show(results):
    for key in sorted(results):
        print("{}: {}".format(key, results[key]))


def _report(model, tag):
    lt = getattr(model, "last_train", None)
    if lt:
        print("INFO: %s: %d steps, %d examples, %.1f examples/sec, last batch loss %.6f" % (
            tag, lt["steps"], lt["examples"], lt["examples"] / max(lt["seconds"], 1e-9), lt["loss"]))


def train_and_eval(model, F):
    for n in range(F.train_epochs):
        print("=" * 30 + " START EPOCH {} ".format(n + 1) + "=" * 30 + "\n")
        for f in list_files(F.train_data):
            t0 = time.time()
            print("<EPOCH {}>: Start training {}".format(n + 1, f))
            model.train(input_fn=lambda: input_fn(f, F.image_train_data, "train", F.batch_size))
            print("<EPOCH {}>: Finish training {}, take {} mins".format(n + 1, f, elapse_time(t0)))
            _report(model, f)
            print("-" * 80)
            t0 = time.time()
            results = model.evaluate(input_fn=lambda: input_fn(F.eval_data, F.image_eval_data, "eval", F.batch_size))
            print("<EPOCH {}>: Finish evaluation {}, take {} mins".format(n + 1, F.eval_data, elapse_time(t0)))
            print("-" * 80)
            _show(results)
        if (n + 1) % F.epochs_per_eval == 0:
            # the reference passes mode 'pred' to evaluate here (train.py:98), which cannot produce labels; the test
            # set is evaluated with its labels
            results = model.evaluate(input_fn=lambda: input_fn(F.test_data, F.image_test_data, "eval", F.batch_size))
            print("<EPOCH {}>: Finish testing {}".format(n + 1, F.test_data))
            print("-" * 80)
            _show(results)


def dynamic_train(model, F):
    data_files = sorted(list_files(F.train_data))
    assert len(data_files) > 1, "Dynamic train mode need more than 1 data file"
    for i in range(len(data_files) - 1):
        train_data, test_data = data_files[i], data_files[i + 1]
        print("=" * 30 + " START TRAINING DATA: {} ".format(train_data) + "=" * 30 + "\n")
        for n in range(F.train_epochs):
            t0 = time.time()
            print("START TRAIN DATA <{}> <EPOCH {}>".format(train_data, n + 1))
            model.train(input_fn=lambda: input_fn(train_data, F.image_train_data, "train", F.batch_size))
            print("FINISH TRAIN DATA <{}> <EPOCH {}> take {} mins".format(train_data, n + 1, elapse_time(t0)))
            _report(model, train_data)
            print("-" * 80)
            t0 = time.time()
            results = model.evaluate(input_fn=lambda: input_fn(test_data, F.image_eval_data, "eval", F.batch_size))
            print("FINISH EVALUATE TEST DATA <{}> <EPOCH {}>: take {} mins".format(test_data, n + 1, elapse_time(t0)))
            print("-" * 80)
            _show(results)


def train_only(model, F):
    for n in range(F.train_epochs):
        print("=" * 30 + " START EPOCH {} ".format(n + 1) + "=" * 30 + "\n")
        for f in list_files(F.train_data):
            t0 = time.time()
            model.train(input_fn=lambda: input_fn(f, F.image_train_data, "train", F.batch_size))
            print("<EPOCH {}>: Finish training {}, take {} mins".format(n + 1, f, elapse_time(t0)))
            _report(model, f)


def main(argv=None):
    conf = Config()
    F, _ = build_parser(conf.train).parse_known_args(argv)
    print("\nModel Type: {}".format(F.model_type))
    model_dir = os.path.join(F.model_dir, F.model_type)
    print("\nModel Directory: {}".format(model_dir))
    print("\nUsing Train Config:")
    for k, v in conf.train.items():
        print("{}: {}".format(k, v))
    print("\nUsing Model Config:")
    for k, v in conf.model.items():
        print("{}: {}".format(k, v))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    rank = int(os.environ.get("RANK", "0"))
    if not F.keep_train and rank == 0:
        shutil.rmtree(model_dir, ignore_errors=True)
        print("Remove model directory: {}".format(model_dir))
    model = build_custom_estimator(model_dir, F.model_type, conf=conf, max_batch=F.batch_size)
    if world > 1 or conf.distribution.get("is_distribution"):
        train_fn = train_only      # "distributed can not including eval" (train.py:213-214)
    elif conf.train["dynamic_train"]:
        train_fn = dynamic_train
        print("Using dynamic train mode.")
    else:
        train_fn = train_and_eval
    train_fn(model, F)


if __name__ == "__main__":
    main()
