#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_fuzz.py -x -q -k "sparse or fused or zipf or multi_hot or row_record or fuzz or full_size" 2>&1 | tail -n 3
python -m pytest tests/test_gpu_fullsize.py -x -q -k "zipf or c4" 2>&1 | tail -n 2
for i in 1 2; do for d in zipf uniform; do echo -n "$d: "; python bench.py --steps 100 --warmup 20 --dist $d --no-pmc --no-cpu-baseline --no-parity | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"])"; done; done
echo -n "c4: "; python bench.py --config c4 --steps 60 --warmup 20 --no-pmc --no-cpu-baseline --no-parity | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"])"
