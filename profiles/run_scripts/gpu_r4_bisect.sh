#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
R='tests/test_gpu_fullsize.py::test_c2_the_graphs_bench_times_are_bit_identical_to_eager_steps_that_match_the_oracle[uniform]'
for F in "$@"; do
  timeout 600 python -X faulthandler -m pytest $F "$R" -q -m gpu -x -p no:cacheprovider > /tmp/bis.txt 2>&1
  echo "== $F + runner: rc $? :: $(grep -c Fatal /tmp/bis.txt) fatal :: $(tail -n 1 /tmp/bis.txt | cut -c1-100)"
done
