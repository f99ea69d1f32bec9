#!/usr/bin/env python
"""`python pred.py` -- prediction entry point of the drop-in path (reference CLI: python/pred.py:24-77) on the MI355X engine.

Flags as the reference: --model_dir --model_type --data_dir (required) --image_data_dir --batch_size --checkpoint_path.
Rows are parsed in 'pred' mode (no label column used), pushed through the forward pass of the HIP engine, and every
prediction is printed the reference's way (pred.py:71-74): the winning class id and its probability."""
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
    p = argparse.ArgumentParser(description="Wide and Deep Model Prediction")
    p.add_argument("--model_dir", type=str, default=cfg["model_dir"], help="Model checkpoint dir for evaluating.")
    p.add_argument("--model_type", type=str, default=cfg["model_type"], help="Valid model types: {'wide', 'deep', 'wide_deep'}.")
    p.add_argument("--data_dir", type=str, help="Prediction data file or dir.")
    p.add_argument("--image_data_dir", type=str, default=None, help="(image tower: out of scope, keep empty)")
    p.add_argument("--batch_size", type=int, default=cfg["batch_size"], help="Number of examples per batch.")
    p.add_argument("--checkpoint_path", type=str, default=cfg.get("checkpoint_path"),
                   help="Path of a specific checkpoint to predict. If None, the latest checkpoint in model_dir is used.")
    return p


def main(argv=None, out=sys.stdout):
    F, _ = build_parser(Config().train).parse_known_args(argv)
    if F.data_dir is None:
        raise ValueError("Must specify prediction data_file by --data_dir")
    print("Model type: {}".format(F.model_type), file=out)
    model_dir = os.path.join(F.model_dir, F.model_type)
    print("Model directory: {}".format(model_dir), file=out)
    model = build_custom_estimator(model_dir, F.model_type, max_batch=F.batch_size)
    print("INFO: " + "=" * 30 + "START PREDICTION" + "=" * 30, file=out)
    t0 = time.time()
    predictions = model.predict(input_fn=lambda: input_fn(F.data_dir, F.image_data_dir, "pred", F.batch_size),
                                predict_keys=None, hooks=None, checkpoint_path=F.checkpoint_path or None)
    n = 0
    for pred_dict in predictions:   # {logits, logistic, probabilities, class_ids, classes}
        class_id = int(pred_dict["class_ids"][0])
        probability = float(pred_dict["probabilities"][class_id])
        print('\nPrediction is "{}" ({:.1f}%)'.format(class_id, 100 * probability), file=out)
        n += 1
    print("INFO: " + "=" * 30 + "FINISH PREDICTION, TAKE {} mins".format(round((time.time() - t0) / 60, 2)) + "=" * 30, file=out)
    return n


if __name__ == "__main__":
    main()
