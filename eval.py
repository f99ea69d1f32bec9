#!/usr/bin/env python
"""`python eval.py` -- evaluation entry point of the drop-in path (reference CLI: python/eval.py:24-93) on the MI355X engine.

Flags and defaults as the reference (conf/train.yaml): --model_dir --model_type --test_data --image_test_data
--batch_size --checkpoint_path.  Restores `<model_dir>/<model_type>` (latest checkpoint unless --checkpoint_path names
one), runs the forward pass of the HIP engine over the test files and prints the canned binary-head metrics sorted by
name, as the reference does (eval.py:88-91)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from wide_deep_amd.build_estimator import build_custom_estimator  # noqa: E402
from wide_deep_amd.dataset import input_fn  # noqa: E402
from wide_deep_amd.read_conf import Config  # noqa: E402


def build_parser(cfg):
    p = argparse.ArgumentParser(description="Evaluate Wide and Deep Model.")
    p.add_argument("--model_dir", type=str, default=cfg["model_dir"], help="Model checkpoint dir for evaluating.")
    p.add_argument("--model_type", type=str, default=cfg["model_type"], help="Valid model types: {'wide', 'deep', 'wide_deep'}.")
    p.add_argument("--test_data", type=str, default=cfg["test_data"], help="Evaluating data dir.")
    p.add_argument("--image_test_data", type=str, default=cfg.get("image_test_data"), help="(image tower: out of scope, keep empty)")
    p.add_argument("--batch_size", type=int, default=cfg["batch_size"], help="Number of examples per batch.")
    p.add_argument("--checkpoint_path", type=str, default=cfg.get("checkpoint_path"),
                   help="Path of a specific checkpoint to evaluate. If None, the latest checkpoint in model_dir is used.")
    return p


def main(argv=None):
    F, _ = build_parser(Config().train).parse_known_args(argv)
    print("Model type: {}".format(F.model_type))
    model_dir = os.path.join(F.model_dir, F.model_type)
    print("Model directory: {}".format(model_dir))
    model = build_custom_estimator(model_dir, F.model_type, max_batch=F.batch_size)
    print("INFO: " + "=" * 30 + " START TESTING" + "=" * 30)
    t0 = time.time()
    results = model.evaluate(input_fn=lambda: input_fn(F.test_data, F.image_test_data or None, "eval", F.batch_size),
                             steps=None, hooks=None, checkpoint_path=F.checkpoint_path or None, name=None)
    print("INFO: " + "=" * 30 + "FINISH TESTING, TAKE {} mins".format(round((time.time() - t0) / 60, 2)) + "=" * 30)
    print("-" * 80)
    for key in sorted(results):
        print("%s: %s" % (key, results[key]))
    return results


if __name__ == "__main__":
    main()
