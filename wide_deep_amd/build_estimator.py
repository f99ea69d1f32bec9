"""conf -> model spec -> Estimator-shaped object (python/lib/build_estimator.py:49-169, 264-294).

`build_model_spec()` is what `_build_model_columns()` does at graph-build time, producing a `plan.ModelSpec` for the
gfx950 engine instead of tf.feature_column objects.  `build_custom_estimator(model_dir, model_type)` keeps the
reference's signature and returns `estimator.WideAndDeepClassifier` (train / evaluate / predict).
"""
import ast

from .plan import CatSlot, CrossKey, DenseCol, ModelSpec, TowerSpec, embedding_dim
from .read_conf import Config

_NORM_KIND = {None: 0, "min_max": 1, "standard": 2, "log": 3}

# tf.train optimizer defaults that matter on this path
_OPT_ALIASES = {
    "Adagrad": "Adagrad", "AdagradOptimizer": "Adagrad", "Ftrl": "Ftrl", "FtrlOptimizer": "Ftrl",
    "Adam": "Adam", "AdamOptimizer": "Adam", "RMSProp": "RMSProp", "RMSPropOptimizer": "RMSProp",
    "SGD": "SGD", "GradientDescentOptimizer": "SGD",
}


_OPT_NAMES = ("Adagrad", "Adam", "Ftrl", "RMSProp", "SGD")
_ACTIVATIONS = ("crelu", "elu", "leaky_relu", "relu", "relu6", "selu", "sigmoid", "softplus", "softsign", "tanh")


def activation_fn(opt):
    """The ten names python/lib/utils/model_util.py:28-59 accepts, same ValueError for anything else.  Nine are built into
    the tower kernels (capi.ACT_IDS); `crelu` (concat(relu(x), relu(-x)): doubles the width every layer hands to the next)
    runs as a relu layer of twice the width with tied halves (plan.FeaturePlan)."""
    if opt not in _ACTIVATIONS:
        raise ValueError("Unsupported activation name: %s. Supported names are: %s" % (opt, _ACTIVATIONS))
    return opt


def _unsupported(opt):
    return ValueError("Unsupported optimizer option: `%s`. Supported names are: %s or an `Optimizer` instance." % (opt, _OPT_NAMES))


def parse_optimizer(opt, default_lr):
    """'Adagrad' | 'tf.train.FtrlOptimizer(learning_rate=0.1, l1_regularization_strength=0.5, ...)' -> (name, kwargs).
    The reference eval()s constructor strings (python/lib/utils/model_util.py:62-105); here the expression is parsed
    with `ast` and only literal arguments are accepted.  Same acceptance rule and error messages for the five names, for
    unknown names / classes and for expressions that are not an optimizer (tests/golden/ref_model_util.json)."""
    if not isinstance(opt, str):
        raise ValueError("optimizer must be a string, got %r" % (opt,))
    text = opt.strip()
    if text in _OPT_NAMES:
        if default_lr is None:
            raise ValueError("learning_rate must be specified when opt is supported string.")
        return text, {"learning_rate": default_lr}
    try:
        node = ast.parse(text, mode="eval").body
    except SyntaxError:
        raise _unsupported(opt)
    if isinstance(node, ast.Name):                      # eval() of an unknown bare name: NameError in the reference
        raise _unsupported(opt)
    if not isinstance(node, ast.Call):                  # evaluates to something that is not an Optimizer
        raise ValueError("The given object is not an Optimizer instance. Given: %s" % text)
    fn = node.func
    cls = fn.attr if isinstance(fn, ast.Attribute) else getattr(fn, "id", None)
    if cls not in _OPT_ALIASES or not cls.endswith("Optimizer"):
        # tf.train has no such class (AttributeError in the reference) -- or it has, and this engine does not build it
        raise _unsupported(opt)
    kwargs = {}
    if node.args:
        kwargs["learning_rate"] = ast.literal_eval(node.args[0])
    for kw in node.keywords:
        kwargs[kw.arg] = ast.literal_eval(kw.value)
    # a constructor without learning_rate: AdamOptimizer has a default (0.001); the others require the argument
    kwargs.setdefault("learning_rate", 0.001 if _OPT_ALIASES[cls] == "Adam" else default_lr)
    return _OPT_ALIASES[cls], kwargs


def opt_tuple(name, kw):
    """(name, kwargs) -> the optimizer tuple of plan.ModelSpec / the oracle, with the tf.train constructor defaults:
       ("SGD", lr) ("Adagrad", lr, initial_accumulator_value) ("Ftrl", lr, l1, l2, initial_accumulator_value)
       ("RMSProp", lr, decay, momentum, epsilon) ("Adam", lr, beta1, beta2, epsilon)
    Non-default variants append elements: ("Ftrl", ..., learning_rate_power) when it is not -0.5, ("Ftrl", ..., learning_rate_power,
    l2_shrinkage_regularization_strength) when the shrinkage is not 0 -- applied in the form TensorFlow >= 1.13 computes and
    its ftrl_test pins ([-0.22578995, -0.44345796] ...): the linear slot takes g + 2 shrinkage var, the accumulator the PLAIN
    g^2.  (The reference only asks for tensorflow >= 1.4; earlier 1.x releases are recalled to square the shrunk gradient into
    the accumulator -- not verifiable here, no TF.  The newer semantics are matched on purpose),
    ("RMSProp", ..., True) for centered=True."""
    lr = float(kw["learning_rate"])
    if name == "SGD":
        return ("SGD", lr)
    if name == "Adagrad":
        return ("Adagrad", lr, float(kw.get("initial_accumulator_value", 0.1)))
    if name == "Ftrl":
        shrink = float(kw.get("l2_shrinkage_regularization_strength", 0.0))
        if shrink < 0.0:        # tf.train.FtrlOptimizer.__init__
            raise ValueError("l2_shrinkage_regularization_strength %f needs to be positive or zero" % shrink)
        lr_power = float(kw.get("learning_rate_power", -0.5))
        if lr_power > 0.0:      # tf.train.FtrlOptimizer.__init__
            raise ValueError("learning_rate_power %f needs to be negative or zero" % lr_power)
        base = ("Ftrl", lr, float(kw.get("l1_regularization_strength", 0.0)),
                float(kw.get("l2_regularization_strength", 0.0)), float(kw.get("initial_accumulator_value", 0.1)))
        if shrink != 0.0:
            return base + (lr_power, shrink)
        return base if lr_power == -0.5 else base + (lr_power,)
    if name == "RMSProp":
        base = ("RMSProp", lr, float(kw.get("decay", 0.9)), float(kw.get("momentum", 0.0)), float(kw.get("epsilon", 1e-10)))
        return base + (True,) if kw.get("centered") else base
    if name == "Adam":
        return ("Adam", lr, float(kw.get("beta1", 0.9)), float(kw.get("beta2", 0.999)), float(kw.get("epsilon", 1e-8)))
    raise ValueError("Unsupported optimizer: %s" % name)


def build_model_spec(conf=None, model_type=None):
    """Feature / cross / model conf -> ModelSpec  (the wiring of python/lib/build_estimator.py:49-169)."""
    conf = conf or Config()
    feature_conf = conf.read_feature_conf()
    cross_conf = conf.read_cross_feature_conf()
    model = conf.model
    train = conf.train
    model_type = model_type or train["model_type"]
    global_dim = model.get("embedding_dim")   # extension, see read_conf.py

    slots, dense = [], []
    for f, c in feature_conf.items():
        kind, trans, param = c["type"], c["transform"], c["parameter"]
        if kind == "category":
            if trans == "hash_bucket":
                dim = int(c.get("embedding_dim") or global_dim or embedding_dim(param))
                slots.append(CatSlot(name=f, kind="hash", num_buckets=int(param), feature=f, deep="embedding", dim=dim))
            elif trans == "vocab":
                vocab = [str(v) for v in param]   # vocabulary_list=map(str, f_param)  (build_estimator.py:103)
                slots.append(CatSlot(name=f, kind="vocab", num_buckets=len(vocab), feature=f, deep="indicator", vocab=vocab))
            else:
                slots.append(CatSlot(name=f, kind="identity", num_buckets=int(param), feature=f, deep="indicator"))
        else:
            norm, bounds = param["normalization"], param["boundaries"]
            p0, p1 = (float(norm[0]), float(norm[1])) if (trans in ("min_max", "standard")) else (0.0, 1.0)
            dense.append(DenseCol(name=f, feature=f, kind=_NORM_KIND[trans], p0=p0, p1=p1))
            if bounds:
                # wide bucketized column wraps the NORMALISED numeric column while the boundaries are raw (quirk C.5)
                slots.append(CatSlot(name=f + "_bucketized", kind="bucket", num_buckets=len(bounds) + 1, feature=f,
                                     deep=None, boundaries=[float(b) for b in bounds],
                                     normalizer=(trans, p0, p1) if trans else None))
    for parts, size, is_deep in cross_conf:
        keys, names = [], []
        for p in parts:
            c = feature_conf[p]
            if c["type"] == "continuous":
                b = [float(v) for v in c["parameter"]["boundaries"]]
                keys.append(CrossKey(p, "bucket", num_buckets=len(b) + 1, boundaries=b))   # un-normalised column (C.5)
                names.append(p + "_bucketized")
            elif c["transform"] == "identity":
                keys.append(CrossKey(p, "identity", num_buckets=int(c["parameter"])))
                names.append(p)
            else:
                keys.append(CrossKey(p, "string"))
                names.append(p)
        nb = int(size)
        if nb < 1:
            raise ValueError("hash_bucket_size must be > 1. hash_bucket_size: %s" % size)
        slots.append(CatSlot(name="_X_".join(sorted(names)), kind="cross", num_buckets=nb, deep="embedding" if is_deep else None,
                             dim=int(global_dim or embedding_dim(size)) if is_deep else 0, cross_keys=keys))

    hidden, mode = model["dnn_hidden_units"], model["dnn_connected_mode"]
    towers = tower_specs(hidden, mode)
    dnn_name, dnn_kw = parse_optimizer(model["dnn_optimizer"], model.get("dnn_initial_learning_rate") or 0.05)
    lin_name, lin_kw = parse_optimizer(model["linear_optimizer"], model.get("linear_initial_learning_rate") or 0.05)
    # learning-rate decay done the way the reference's comment describes it (python/lib/joint.py:65-66, 145-154), behind a flag:
    # the reference itself never decays (quirk C.2).  Only optimizers given by name take the model_fn's learning rate.
    lr_decay = None
    if train.get("lr_decay"):
        # decay_steps (python/lib/joint.py:78 `_num_examples / _batch_size`): true division -- joint.py:25 has
        # `from __future__ import division`, so the quotient is a float under the reference's Python 2 as well
        steps = float(train["num_examples"]) / float(train["batch_size"])
        lr_decay = {}
        for scope, key, rk in (("dnn", "dnn_optimizer", "dnn_decay_rate"), ("linear", "linear_optimizer", "linear_decay_rate")):
            rate = model.get(rk) or 1
            if str(model[key]).strip() in _OPT_NAMES and float(rate) != 1.0:
                lr_decay[scope] = (float(rate), steps)
    # weight column is switched on when EITHER weight is set (build_estimator.py:43-46) ...
    use_w = train["pos_sample_loss_weight"] is not None or train["neg_sample_loss_weight"] is not None
    return ModelSpec(model_type=model_type, slots=slots, dense_cols=dense, towers=towers,
                     activation=activation_fn(model.get("dnn_activation_function")),
                     batch_norm=bool(model.get("dnn_batch_normalization")), dropout=model.get("dnn_dropout") or None,
                     dnn_opt=opt_tuple(dnn_name, dnn_kw), lin_opt=opt_tuple(lin_name, lin_kw),
                     use_weight_column=use_w, pos_weight=float(train["pos_sample_loss_weight"] or 1.0),
                     neg_weight=float(train["neg_sample_loss_weight"] or 1.0), lr_decay=lr_decay or None)


def tower_specs(hidden, mode):
    """dnn_hidden_units / dnn_connected_mode -> one TowerSpec per DNN (python/lib/dnn.py:237-258): a 1-D hidden list is one
    tower; ONE mode -- a name, or a connection list, recognised as the reference does by a first item of three characters
    ('0-1') -- serves every tower, anything else is one mode per tower."""
    multi = bool(hidden) and isinstance(hidden[0], (list, tuple))
    hidden = [list(h) for h in hidden] if multi else [list(hidden)]
    from .plan import is_connection_list
    if not isinstance(mode, str) and len(mode) == 0:
        raise ValueError("dnn_connected_mode is empty")
    # one connection list for a single tower whatever the length of its items ('0-10'); for several towers the reference's
    # three-character test (python/lib/dnn.py:253-258) decides between "one list for all" and "one mode per tower"
    one = (isinstance(mode, str) or (len(mode) > 0 and isinstance(mode[0], str) and len(mode[0]) == 3)
           or (not multi and is_connection_list(mode)))
    modes = [mode] * len(hidden) if one else list(mode)
    if len(modes) != len(hidden):
        raise ValueError("dnn_connected_mode lists %d modes for %d DNNs (dnn_hidden_units)" % (len(modes), len(hidden)))
    return [TowerSpec([int(h) for h in hs], _mode_name(m, len(hs))) for hs, m in zip(hidden, modes)]


def _mode_name(m, n_hidden):
    """Connected mode of one tower: a name of python/lib/dnn.py:58-66 or a connection list ['0-1', '0-3', '1-2'] (also as one
    string '0-1,0-3,1-2': the conf reader only lets strings through) -> tuple of (i, j) pairs."""
    from .plan import is_connection_list, parse_connections
    if is_connection_list(m):
        return parse_connections(m, n_hidden)
    return {"normal": "simple"}.get(m, m)


def build_custom_estimator(model_dir, model_type, conf=None, **engine_kw):
    """Reference signature (python/lib/build_estimator.py:264-294)."""
    from .estimator import WideAndDeepClassifier
    conf = conf or Config()
    spec = build_model_spec(conf, model_type)
    return WideAndDeepClassifier(spec, model_dir=model_dir, runconfig=conf.runconfig, **engine_kw)


def build_estimator(model_dir, model_type):
    raise NotImplementedError("the canned tf.estimator builders (python/lib/build_estimator.py:201-261) are out of scope: "
                              "train.py uses build_custom_estimator (SURVEY section 2, row 12)")
