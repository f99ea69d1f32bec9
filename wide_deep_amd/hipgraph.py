"""Lifetime of captured hipGraphs (torch.cuda.CUDAGraph) on ROCm 7.2.

A graph executable that is DESTROYED while others are alive leaves the runtime with dangling stream bookkeeping: a later
`hipGraphLaunch` of a live, multi-branch graph segfaults in `hip::Graph::UpdateStreams` <- `hip::GraphExec::Run`
(profiles/r5_hipgraph_segv_backtrace.txt: the whole GPU suite in one process, the chained step graphs replayed behind ~45 tests
whose graphs the garbage collector had destroyed; with every graph of the process kept alive the same run passes,
profiles/r5_hipgraph_segv_pinned_graphs_pass.txt; scripts/hipgraph_segv_repro.sh reproduces both).  Nothing of ours is
referenced by a dead graph -- events, side streams and buffers of a capture only matter while it is replayed -- so the cure is
on the lifetime side: every graph this package captures is registered here and lives as long as the process.  A graph
executable is a few kernel-argument blocks (no step of ours allocates during capture: the private pool of a graph stays
empty); a training process captures a handful (python/train.py:65-165 runs one loop for its whole life).

Growth is bounded by what the owners capture, not here: the Estimator caches one graph per (mode, batch size) and bench.py / the
tests a fixed set.  A process that re-captures without end (per batch shape, per epoch, after every `import_state`) grows by one
executable per capture; `live_graphs()` reports the count, a warning is logged once it passes WD_GRAPH_LIVE_WARN (default 512),
and `release_all()` is the one safe way down: dropping EVERY graph of the process at a point where none will be replayed again
(what breaks a replay is a destroyed executable beside a live multi-branch one, so nothing may stay live)."""
import os
import warnings

import torch

_LIVE = []
_WARNED = False


def new_graph():
    """torch.cuda.CUDAGraph() that is never destroyed before the process ends (WD_GRAPH_KEEPALIVE=0: an ordinary graph object,
    destroyed with its owner -- what scripts/hipgraph_segv_repro.sh sets to show the crash)."""
    global _WARNED
    g = torch.cuda.CUDAGraph()
    if os.environ.get("WD_GRAPH_KEEPALIVE", "1") != "0":
        _LIVE.append(g)
        if not _WARNED and len(_LIVE) > int(os.environ.get("WD_GRAPH_LIVE_WARN", "512")):
            _WARNED = True
            warnings.warn("wide_deep_amd.hipgraph: %d captured graphs are pinned for the life of the process (ROCm 7.2 crashes "
                          "replays of live graphs once others are destroyed); cache captures per batch shape instead of "
                          "re-capturing, or call hipgraph.release_all() when no graph will be replayed again" % len(_LIVE))
    return g


def live_graphs():
    """number of graphs pinned so far (monotonic between release_all() calls)"""
    return len(_LIVE)


def release_all():
    """Drop the pins of ALL graphs captured so far.  Only safe when the caller will not replay any of them again (their owners
    may then let them be destroyed): call it behind a `torch.cuda.synchronize()` at the end of a phase whose graphs are all
    discarded together.  Returns how many were released."""
    global _WARNED
    torch.cuda.synchronize()
    n = len(_LIVE)
    del _LIVE[:]
    _WARNED = False
    return n
