#!/usr/bin/env python
"""Regenerate conf/*.yaml (this repo's default configuration, in the reference's YAML schema) from the
reference's shipped defaults, so that `BASELINE configs[0]` ("repo default conf/") means the same model here.

Only the VALUES are carried over (they are the configuration the hot path is specified on); the files are re-emitted
in compact flow style with this repo's own comments.  Run in the build container (needs /root/reference):
    python scripts/make_default_conf.py
"""
import os
import sys

import yaml

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/conf"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")


def load(name):
    with open(os.path.join(REF, name)) as f:
        return yaml.safe_load(f)


def flow(v):
    if v is None:
        return "~"
    out = yaml.safe_dump(v, default_flow_style=True, width=10 ** 6).strip()
    return out.splitlines()[0] if out.endswith("...") else out     # scalars carry a document-end marker


def main():
    os.makedirs(OUT, exist_ok=True)
    # schema: column position -> field name (61-field TSV; column 1 is the label `clk`)
    schema = load("schema.yaml")
    with open(os.path.join(OUT, "schema.yaml"), "w") as f:
        f.write("# TSV column position -> field name (one flow mapping, listed by field name; column 1 `clk` is the label)\n{\n")
        items = sorted(schema.items(), key=lambda kv: str(kv[1]))     # by field name; a mapping has no order
        for i in range(0, len(items), 5):
            f.write("  " + ", ".join("%d: %s" % (k, v) for k, v in items[i:i + 5]) + ",\n")
        f.write("}\n")
    # features: one line per used feature
    feat = load("feature.yaml")
    with open(os.path.join(OUT, "feature.yaml"), "w") as f:
        f.write("# <feature>: {type: category|continuous, transform: hash_bucket|vocab|identity|min_max|standard|log,\n"
                "#             parameter: bucket count | vocabulary list | {normalization: [a, b], boundaries: [...]}}\n"
                "# Fields of schema.yaml that are not listed here are ignored.  Optional extension (not in the reference):\n"
                "#   embedding_dim: N   overrides the reference's 2**ceil(ln(buckets**0.25)) rule for this feature.\n")
        for k, v in feat.items():
            f.write("%s: %s\n" % (k, flow(v)))
    cross = load("cross_feature.yaml")
    with open(os.path.join(OUT, "cross_feature.yaml"), "w") as f:
        f.write("# a&b[&c]: {hash_bucket_size: <in THOUSANDS of buckets>, is_deep: 0|1 (also embed the cross in the tower)}\n"
                "# (sorted by name; the order of the entries has no meaning)\n")
        for k in sorted(cross):
            f.write("%s: %s\n" % (k, flow(cross[k])))
    model = load("model.yaml")
    with open(os.path.join(OUT, "model.yaml"), "w") as f:
        f.write("# wide (linear) side, deep (dnn) side; cnn_* keys are accepted and ignored (image tower is out of scope)\n")
        lin = {k: v for k, v in model.items() if k.startswith("linear_")}
        dnn = {k: v for k, v in model.items() if k.startswith("dnn_")}
        cnn = {k: v for k, v in model.items() if k.startswith("cnn_")}
        rest = {k: v for k, v in model.items() if k not in lin and k not in dnn and k not in cnn}
        for grp in (lin, dnn, cnn, rest):
            for k in sorted(grp):
                v = grp[k]
                f.write("%s: %s\n" % (k, flow(v) if isinstance(v, (list, dict)) else ("~" if v is None else flow(v))))
    train = load("train.yaml")
    with open(os.path.join(OUT, "train.yaml"), "w") as f:
        f.write("# train: run control; distribution: kept for schema compatibility (multi-GPU here = one process per GPU via\n"
                "# torch.distributed, see DESIGN.md section 6); runconfig: checkpoint cadence\n")
        for sec in sorted(train):
            f.write("%s: {\n" % sec)
            for k in sorted(train[sec]):
                f.write("  %s: %s,\n" % (k, flow(train[sec][k])))
            f.write("}\n")


if __name__ == "__main__":
    main()
