"""CPU: train-op wiring against golden events recorded while EXECUTING the reference's `_wide_deep_combined_model_fn`
(python/lib/joint.py:81-269) on a stub tensorflow (tests/golden/make_ref_joint_golden.py -> tests/golden/ref_joint.json):
optimizer per scope, the inert learning-rate decay (quirk C.2), global-step increments per batch (quirk C.4), logits sum."""
import json
import os
import re

from wide_deep_amd import build_estimator as BE
from wide_deep_amd.read_conf import Config

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_joint.json")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _steps_per_batch(events):
    n = sum(1 for e in events if e["event"] == "minimize" and e["global_step"] == {"op": "the_global_step"})
    return n + sum(e["value"] for e in events if e["event"] == "assign_add" and e["target"] == {"op": "the_global_step"})


def test_global_step_advances_like_the_reference_train_op():
    exp = {mt: _steps_per_batch(ev) for mt, ev in G["model_types"].items()}
    assert exp == {"wide": 2, "deep": 2, "wide_deep": 3}
    # the engine's own accounting (wide_deep_amd/engine.py) and the CPU stand-in of tests/test_estimator_host_cpu.py
    src = open(os.path.join(ROOT, "wide_deep_amd", "engine.py")).read()
    incs = re.findall(r'self\.global_step \+= (\d) if self\.spec\.model_type == "wide_deep" else (\d)', src)
    assert incs and all(i == ("3", "2") for i in incs)


def test_optimizer_per_scope_and_inert_learning_rate_decay():
    spec = BE.build_model_spec(Config(), "wide_deep")
    mins = [e for e in G["model_types"]["wide_deep"] if e["event"] == "minimize"]
    by_scope = {m["var_list"]["scope"]: m["optimizer"] for m in mins}
    assert sorted(by_scope) == ["dnn", "linear"]                       # TRAINABLE_VARIABLES filtered by parent scope
    dnn, lin = by_scope["dnn"], by_scope["linear"]
    # dnn scope: Adagrad whose learning rate is exponential_decay over a FRESH Variable(0) nobody increments -> constant
    lr = dnn["kwargs"]["learning_rate"]
    assert dnn["class"] == "AdagradOptimizer" and lr["op"] == "exponential_decay" and lr["staircase"] is False
    assert lr["global_step"] == {"op": "fresh_variable", "initial_value": 0}
    assert lr["global_step"] != {"op": "the_global_step"}
    effective = lr["initial"] * lr["decay_rate"] ** (0 / lr["decay_steps"])
    assert spec.dnn_opt == ("Adagrad", effective, 0.1) and effective == G["constants"]["dnn_init_learning_rate"]
    # linear scope: the constructor string of conf/model.yaml carries its own learning rate; the decayed one is not used
    assert lin["class"] == "FtrlOptimizer" and lin["kwargs"]["learning_rate"] == 0.1
    assert spec.lin_opt == ("Ftrl", 0.1, lin["kwargs"]["l1_regularization_strength"], float(lin["kwargs"]["l2_regularization_strength"]), 0.1)
    assert G["constants"]["linear_decay_rate"] == 0.8 and G["constants"]["dnn_decay_rate"] == 0.8      # configured, inert


def test_logits_are_summed_and_linear_model_uses_sum_combiner():
    ev = G["model_types"]["wide_deep"]
    final = [e for e in ev if e["event"] == "add_n"][-1]
    assert [i["op"] for i in final["inputs"]] == ["logits_sum", "linear_logits"]        # dnn (sum over towers) + linear
    lm = [e for e in ev if e["event"] == "linear_model"][0]["kwargs"]
    assert lm["sparse_combiner"] == "sum" and lm["units"] == 1
    assert [e["event"] for e in G["model_types"]["wide"]].count("minimize") == 1
    assert [e["event"] for e in G["model_types"]["deep"]].count("minimize") == 1
