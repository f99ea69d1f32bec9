"""Configuration reader with the reference's interface (python/lib/read_conf.py:21-279).

Same five YAML files, same `Config` attribute names (`config / train / distribution / runconfig / model`,
`read_schema()`, `read_feature_conf()`, `read_cross_feature_conf()`, `get_feature_name(kind)`), same validation
outcomes (exception types) and the same quirks (cross `hash_bucket_size` is in THOUSANDS, an empty
value raises; `is_deep` defaults to 1; every property access re-reads the file).

Where the files live: `$WD_CONF_DIR`, else `<repo>/conf` (the reference resolves `../../conf` relative to its own
package, python/lib/read_conf.py:11).  A reference checkout's conf/ directory can be used unchanged.

Extension (opt-in, absent from the reference): a feature may carry `embedding_dim: N` and model.yaml may carry
`embedding_dim: N` to override the reference's derived embedding width (BASELINE configs 2-5 fix it to 16 / 64).
"""
import os

import yaml

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONF_DIR = os.path.join(os.path.dirname(_PKG_DIR), "conf")

_FILES = {
    "schema": "schema.yaml",
    "data_process": "data_process.yaml",
    "feature": "feature.yaml",
    "cross_feature": "cross_feature.yaml",
    "model": "model.yaml",
    "train": "train.yaml",
    "serving": "serving.yaml",
}

_CATEGORY_TRANSFORMS = ("hash_bucket", "identity", "vocab")
_CONTINUOUS_TRANSFORMS = ("min_max", "log", "standard")


def conf_dir():
    return os.environ.get("WD_CONF_DIR") or DEFAULT_CONF_DIR


def _load(path):
    with open(path) as f:
        return yaml.safe_load(f)


def _is_num(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


class Config(object):
    """Same constructor keywords as the reference (file NAMES inside the conf directory)."""

    def __init__(self, schema_conf_file=_FILES["schema"], data_process_conf_file=_FILES["data_process"],
                 feature_conf_file=_FILES["feature"], cross_feature_conf_file=_FILES["cross_feature"],
                 model_conf_file=_FILES["model"], train_conf_file=_FILES["train"],
                 serving_conf_file=_FILES["serving"], base_dir=None):
        base = base_dir or conf_dir()
        self._paths = {
            "schema": os.path.join(base, schema_conf_file),
            "data_process": os.path.join(base, data_process_conf_file),
            "feature": os.path.join(base, feature_conf_file),
            "cross_feature": os.path.join(base, cross_feature_conf_file),
            "model": os.path.join(base, model_conf_file),
            "train": os.path.join(base, train_conf_file),
            "serving": os.path.join(base, serving_conf_file),
        }

    # ---- schema / features --------------------------------------------------------------------------
    def read_schema(self):
        """{column position (1-based): lower-cased field name}  (python/lib/read_conf.py:41-43)."""
        return {pos: str(name).lower() for pos, name in _load(self._paths["schema"]).items()}

    def read_data_process_conf(self):
        return _load(self._paths["data_process"])

    @staticmethod
    def _check_feature(name, known, entry):
        kind, trans, param = entry.get("type"), entry.get("transform"), entry.get("parameter")
        if kind is None:
            raise ValueError("Type are required in feature conf, found empty value for feature `%s`" % name)
        if name not in known:
            raise ValueError("Invalid feature name `%s` in feature conf, must be consistent with schema conf" % name)
        assert kind in ("category", "continuous"), (
            "Invalid type `%s` for feature `%s` in feature conf, must be 'category' or 'continuous'" % (kind, name))
        if kind == "category":
            assert trans in _CATEGORY_TRANSFORMS, (
                "Invalid transform `%s` for feature `%s` in feature conf, must be one of `hash_bucket`, `vocab`, "
                "`identity`." % (trans, name))
            if trans in ("hash_bucket", "identity"):
                if not isinstance(param, int) or isinstance(param, bool):
                    raise TypeError("Invalid parameter `%s` for feature `%s` in feature conf, %s parameter must be an "
                                    "integer." % (param, name, trans))
            elif not isinstance(param, (list, tuple)):
                raise TypeError("Invalid parameter `%s` for feature `%s` in feature conf, vocab parameter must be a "
                                "list." % (param, name))
            return
        norm, bounds = param["normalization"], param["boundaries"]
        if trans:
            assert trans in _CONTINUOUS_TRANSFORMS, (
                "Invalid transform `%s` for feature `%s` in feature conf, continuous feature transform must be "
                "`min_max` or `log` or `standard`." % (trans, name))
            # the reference's `trans == 'min_max' or 'standard'` is always true: a 2-list is demanded for `log` too
            if not isinstance(norm, (list, tuple)) or len(norm) != 2:
                raise TypeError("Invalid normalization parameter `%s` for feature `%s` in feature conf, must be 2 "
                                "elements list for `min_max` or `standard` scaler." % (norm, name))
            if trans == "min_max":
                lo, hi = norm
                if not _is_num(lo) or not _is_num(hi):
                    raise TypeError("Invalid normalization parameter `%s` for feature `%s` in feature conf, list "
                                    "elements must be int or float." % (norm, name))
                assert lo < hi, ("Invalid normalization parameter `%s` for feature `%s` in feature conf, [min, max] "
                                 "list elements must be min<max" % (norm, name))
            elif trans == "standard":
                mean, std = norm
                if not _is_num(mean):
                    raise TypeError("Invalid normalization parameter `%s` for feature `%s` in feature conf, "
                                    "parameter mean must be int or float." % (mean, name))
                if not _is_num(std) or std <= 0:
                    raise TypeError("Invalid normalization parameter `%s` for feature `%s` in feature conf, "
                                    "parameter std must be a positive number." % (std, name))
        if bounds:
            if not isinstance(bounds, (list, tuple)):
                raise TypeError("Invalid parameter `%s` for feature `%s` in feature conf, discretize parameter must "
                                "be a list." % (bounds, name))
            for v in bounds:
                assert _is_num(v), ("Invalid parameter `%s` for feature `%s` in feature conf, discretize parameter "
                                    "element must be integer or float." % (bounds, name))

    def read_feature_conf(self):
        """{feature: {type, transform, parameter}} for the USED features, validated."""
        conf = _load(self._paths["feature"])
        known = set(self.read_schema().values())
        for name, entry in conf.items():
            self._check_feature(str(name).lower(), known, entry)
        return conf

    @staticmethod
    def _check_cross(name, feature_conf, entry):
        parts = [p.strip() for p in name.split("&")]
        size, is_deep = entry.get("hash_bucket_size"), entry.get("is_deep")
        assert len(parts) > 1, "Invalid cross feature name `%s` in cross feature conf,at least 2 features" % name
        for p in parts:
            if p not in feature_conf:
                raise ValueError("Invalid cross feature name `%s` in cross feature conf, must be consistent with "
                                 "feature conf" % name)
            if feature_conf[p]["type"] == "continuous":
                assert feature_conf[p]["parameter"]["boundaries"] is not None, (
                    "Continuous feature must be set bounaries to be bucketized in feature conf as cross feature")
        if size:
            assert _is_num(size), ("Invalid hash_bucket_size `%s` for features `%s` in cross feature conf, expected "
                                   "int or float" % (size, name))
        if is_deep:
            assert is_deep in (0, 1), "Invalid is_deep `%s` for features `%s`, expected 0 or 1." % (is_deep, name)

    def read_cross_feature_conf(self):
        """[(feature list, hash_bucket_size, is_deep)].  hash_bucket_size is the conf value x 1000 (may be a float,
        e.g. 0.1 -> 100.0); 0 -> 10000; an EMPTY value raises TypeError (quirk C.7); is_deep defaults to 1."""
        conf = _load(self._paths["cross_feature"])
        feature_conf = self.read_feature_conf()
        out = []
        for name, entry in conf.items():
            self._check_cross(name, feature_conf, entry)
            parts = [p.strip() for p in name.split("&")]
            size = 1000 * entry["hash_bucket_size"] or 10000
            deep = entry["is_deep"] if entry.get("is_deep") is not None else 1
            out.append((parts, size, deep))
        return out

    # ---- model / train ------------------------------------------------------------------------------
    @staticmethod
    def _need(key, v):
        if v is None:
            raise ValueError("Required type for key `%s`, found None." % key)

    @staticmethod
    def _need_str(key, v):
        if not isinstance(v, str):
            raise ValueError("String type is required for key `%s`, found `%s`." % (key, v))

    @staticmethod
    def _need_num(key, v):
        if not isinstance(v, (int, float)):
            raise ValueError("Numeric type is required for key `%s`, found `%s`." % (key, v))

    @staticmethod
    def _need_bool(key, v):
        if v not in (True, False, 1, 0):
            raise ValueError("Bool type is required for key `%s`, found `%s`." % (key, v))

    @staticmethod
    def _need_list(key, v):
        if not isinstance(v, (list, tuple)):
            raise ValueError("List type is required for key `%s`, found `%s`." % (key, v))

    def _read_model_conf(self):
        conf = _load(self._paths["model"])
        # the reference's list misses a comma, fusing 'dnn_activation_function' with 'cnn_optimizer'
        # (python/lib/read_conf.py:183-184): those two keys are therefore NOT type-checked there either
        required_str = ("linear_optimizer", "dnn_optimizer", "dnn_connected_mode")
        optional_num = ("linear_initial_learning_rate", "linear_decay_rate", "dnn_initial_learning_rate",
                        "dnn_decay_rate", "dnn_l1", "dnn_l2")
        optional_bool = ("dnn_batch_normalization", "cnn_use_flag")
        for k, v in conf.items():
            if k in required_str:
                self._need(k, v)
                self._need_str(k, v)
            elif k in optional_num:
                if v:
                    self._need_num(k, v)
            elif k in optional_bool:
                if v:
                    self._need_bool(k, v)
            elif k == "dnn_hidden_units":
                self._need(k, v)
                self._need_list(k, v)
        return conf

    def _read_train_conf(self):
        conf = _load(self._paths["train"])
        required_str = ("model_dir", "model_type", "train_data", "test_data")
        required_num = ("train_epochs", "epochs_per_eval", "batch_size", "num_examples")
        optional_num = ("pos_sample_loss_weight", "neg_sample_loss_weight", "num_parallel_calls")
        required_bool = ("keep_train", "multivalue", "dynamic_train")
        for k, v in conf["train"].items():
            if k in required_str:
                self._need(k, v)
                self._need_str(k, v)
            elif k in required_num:
                self._need(k, v)
                self._need_num(k, v)
            elif k in optional_num:
                if v:
                    self._need_num(k, v)
            elif k in required_bool:
                self._need(k, v)
                self._need_bool(k, v)
        return conf

    @property
    def config(self):
        return self._read_train_conf()

    @property
    def train(self):
        return self._read_train_conf()["train"]

    @property
    def distribution(self):
        return self._read_train_conf()["distribution"]

    @property
    def runconfig(self):
        return self._read_train_conf()["runconfig"]

    @property
    def model(self):
        return self._read_model_conf()

    @property
    def serving(self):
        return _load(self._paths["serving"])

    def get_feature_name(self, feature_type="all"):
        """'all' (schema order, label removed) | 'used' | 'unused' | 'category' | 'continuous'."""
        feature_conf = self.read_feature_conf()
        schema = self.read_schema()
        names = [schema[k] for k in sorted(schema)]
        names.remove("clk")
        if feature_type == "all":
            return names
        if feature_type == "used":
            return list(feature_conf.keys())
        if feature_type == "unused":
            return set(names) - set(feature_conf.keys())
        if feature_type in ("category", "continuous"):
            return [f for f, c in feature_conf.items() if c["type"] == feature_type]
        raise ValueError("Invalid parameter, must be one of 'all', 'used', 'category, 'continuous")
