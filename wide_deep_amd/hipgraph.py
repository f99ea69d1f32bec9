"""Lifetime of captured hipGraphs (torch.cuda.CUDAGraph) on ROCm 7.2.

A graph executable that is DESTROYED while others are alive leaves the runtime with dangling stream bookkeeping: a later
`hipGraphLaunch` of a live, multi-branch graph segfaults in `hip::Graph::UpdateStreams` <- `hip::GraphExec::Run`
(profiles/r5_hipgraph_segv_backtrace.txt: the whole GPU suite in one process, the chained step graphs replayed behind ~45 tests
whose graphs the garbage collector had destroyed; with every graph of the process kept alive the same run passes,
profiles/r5_hipgraph_segv_pinned_graphs_pass.txt; scripts/hipgraph_segv_repro.sh reproduces both).  Nothing of ours is
referenced by a dead graph -- events, side streams and buffers of a capture only matter while it is replayed -- so the cure is
on the lifetime side: every graph this package captures is registered here and lives as long as the process.  A graph
executable is a few kernel-argument blocks (no step of ours allocates during capture: the private pool of a graph stays
empty); a training process captures a handful (python/train.py:65-165 runs one loop for its whole life)."""
import os

import torch

_LIVE = []


def new_graph():
    """torch.cuda.CUDAGraph() that is never destroyed before the process ends (WD_GRAPH_KEEPALIVE=0: an ordinary graph object,
    destroyed with its owner -- what scripts/hipgraph_segv_repro.sh sets to show the crash)."""
    g = torch.cuda.CUDAGraph()
    if os.environ.get("WD_GRAPH_KEEPALIVE", "1") != "0":
        _LIVE.append(g)
    return g


def live_graphs():
    return len(_LIVE)
