#!/bin/bash
# The hipGraphLaunch segfault of ROCm 7.2 behind destroyed graph executables (wide_deep_amd/hipgraph.py):
# the whole GPU suite in ONE process under rocgdb; on SIGSEGV the native backtrace of every thread goes to $OUT/gdb.txt.
#   WD_GRAPH_KEEPALIVE=0 bash scripts/hipgraph_segv_repro.sh    graphs die with their owners (rounds 1-4): SIGSEGV in
#                                                               hip::Graph::UpdateStreams at a replay of the chained step graphs
#                                                               (tests/test_gpu_fullsize.py::test_c2_the_graphs_bench_times_...)
#   bash scripts/hipgraph_segv_repro.sh                         every captured graph kept alive (the default): the suite passes
# Evidence of round 5: profiles/r5_hipgraph_segv_backtrace.txt, profiles/r5_hipgraph_segv_pinned_graphs_pass.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/segv; mkdir -p $OUT
export PYTHONFAULTHANDLER=1
cat > $OUT/cmds.gdb <<'G'
set pagination off
set confirm off
handle SIGUSR1 nostop noprint pass
handle SIG34 nostop noprint pass
handle SIG35 nostop noprint pass
handle SIG36 nostop noprint pass
run
echo \n==== signal / exit ====\n
info signal SIGSEGV
bt 40
echo \n==== all threads ====\n
thread apply all bt 12
info sharedlibrary amdhip
quit
G
timeout ${SEGV_TIMEOUT:-1500} rocgdb -q -batch -x $OUT/cmds.gdb --args python -X faulthandler -m pytest tests -q -m gpu -x -p no:cacheprovider "$@" > $OUT/gdb.txt 2>&1
echo "rocgdb exit $?"; grep -n "SIGSEGV\|passed\|failed\|Segmentation\|exited" $OUT/gdb.txt | head; grep -n "==== signal" -A60 $OUT/gdb.txt | head -90
