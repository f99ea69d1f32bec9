#!/bin/bash
# wavefront priority of the row update beside the products (WD_UPDATE_PRIO), now that the dense tail is 6 us: the update's end
# (plus the cross-queue join) gates the next tower
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r4updprio}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
B="--no-cpu-baseline --no-pmc --no-parity"
line() { python - "$@" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %.4f ms/step %s" % (sys.argv[2], d["ms_per_step"], d.get("repeats_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2; do for m in 0 3 1; do
  WD_UPDATE_PRIO=$m timeout 300 python bench.py --steps 20 --warmup 5 $B > $OUT/c2_p${m}_$i.json 2>> $OUT/err.txt; line $OUT/c2_p${m}_$i.json "C2 update prio $m run $i"
done; done
WD_UPDATE_PRIO=3 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python bench.py --steps 60 --warmup 10 --pool 16 --repeats 1 $B > $OUT/prof.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace*.csv" | head -1)
python scripts/trace_window.py $T k_tower_chain 45 2 > $OUT/step_timeline.txt; grep -v hash_bucket $OUT/step_timeline.txt | cut -c1-110 | head -18
rm -rf $OUT/prof
tail -n 3 $OUT/err.txt
