#!/bin/bash
# round 5, call 13: what each side kernel costs the tower: step timelines with the sort / the gather of the next batch left out
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_call13; mkdir -p $OUT
i=0
for v in "WD_DIAG_SKIP=" "WD_DIAG_SKIP=sort" "WD_DIAG_SKIP=gather" "WD_DIAG_SKIP=sort,gather"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 40 --warmup 10 --pool 16 --repeats 1 --no-cpu-baseline --no-pmc --no-parity > $OUT/prof$i.log 2>&1
  T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
  echo "== $v"; grep -o '"ms_per_step": [0-9.]*' $OUT/prof$i.log | head -1
  python scripts/trace_window.py $T k_tower_chain 36 1 > $OUT/timeline$i.txt; grep -v "hash_bucket" $OUT/timeline$i.txt | head -10 | tail -9
  rm -rf $OUT/prof
done 2>&1 | tee $OUT/side.txt
