"""Replay of consecutive train steps as ONE hipGraph (the outer loop of python/train.py:65-165 over resident batches).

A train step on a resident batch is a fixed sequence of launches on fixed buffers (engine.py allocates nothing per step),
so `n` consecutive steps -- each on its OWN batch: raw tokens -> wd_hash_bucket -> forward -> loss -> backward -> both
optimizers -- can be captured once and replayed; the ~11 us between two graph launches is then paid once per `n` steps.
Used by bench.py (the timed path) and by tests/test_gpu_fullsize.py (the same path against the CPU oracle).
"""
import torch

from . import synth


def step_eager(eng, tb, ids_input=False):
    """One train step on a resident TokenBatch: hash in step unless the ids are already there."""
    bt = tb.batch if ids_input else synth.hash_tokens(eng, tb)
    return eng.train_step(bt)


class StepGraph:
    """`len(token_batches)` consecutive train steps in one hipGraph.  The engine must have run at least one eager step
    on a side stream before (lazy allocations / module loads are not capturable)."""

    def __init__(self, eng, token_batches, ids_input=False, stream=None):
        self.eng = eng
        self.n = len(token_batches)
        self.stream = stream or torch.cuda.Stream()
        self.graph = torch.cuda.CUDAGraph()
        self.stream.wait_stream(torch.cuda.current_stream())
        gs = eng.global_step            # capturing executes nothing: the counter must not move
        with torch.cuda.graph(self.graph, stream=self.stream):
            for tb in token_batches:
                step_eager(eng, tb, ids_input)
        eng.global_step = gs
        self._bump = (3 if eng.spec.model_type == "wide_deep" else 2) * self.n

    def replay(self):
        self.graph.replay()
        self.eng.global_step += self._bump
        return self.eng.loss


def warm(eng, token_batches, ids_input=False, steps=2):
    """Eager steps on a side stream (what torch needs before a capture); returns that stream."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(steps):
            step_eager(eng, token_batches[i % len(token_batches)], ids_input)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return side
