import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wide_deep_amd import synth, pipeline
from wide_deep_amd.engine import WideDeepEngine
from wide_deep_amd.plan import criteo_spec
B = 8192
spec = criteo_spec()
eng = WideDeepEngine(spec, max_batch=B, max_nnz=B * 26 * 4, seed=0)
hbs = [synth.make_raw_batch(eng.plan, B, seed=1 + i) for i in range(4)]
tbs = [synth.TokenBatch(eng.plan, hb) for hb in hbs]
side = pipeline.warm(eng, tbs)
print("warm ok, folded", eng._folded, flush=True)
for n in (1, 2, 3, 4):
    print("capturing", n, flush=True)
    g = pipeline.StepGraph(eng, tbs[:n], stream=side)
    print("captured", n, "pipelined", g.pipelined, flush=True)
    g.replay(); torch.cuda.synchronize()
    print("replayed", n, float(eng.loss), flush=True)
