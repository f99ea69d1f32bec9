#!/bin/bash
# tower kernel: parity tests + stand-alone timings (scripts/bench_chain.py); optional second library for an A/B (WD_HIP_LIB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -n 2
for i in 1 2; do python scripts/bench_chain.py 2>&1 | grep -v amdgpu | head -4; done
if [ -n "${OTHER_LIB:-}" ]; then echo "--- $OTHER_LIB"; for i in 1 2; do WD_HIP_LIB=$OTHER_LIB python scripts/bench_chain.py 2>&1 | grep -v amdgpu | head -4; done; fi
