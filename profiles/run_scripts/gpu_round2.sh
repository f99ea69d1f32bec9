#!/bin/bash
# round-2 evidence: bench lines of every config, kernel stats + step timeline of the default bench command, tower stage stamps,
# fp16 GEMM microbench + PMC.  Output -> gpurun_out/<tag>/ ; what is judged is copied into profiles/ (r2z_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
TAG=${1:-r2z}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ "${PYTEST:-0}" = 1 ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -n 3 $OUT/pytest_gpu.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -n 1 $OUT/smoke.txt
fi
timeout 600 python bench.py > $OUT/bench_c2_uniform.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench_c2_uniform.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2_uniform_driver_args.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c2_uniform_driver_args.json
timeout 300 python bench.py --dist zipf --no-cpu-baseline --no-pmc > $OUT/bench_c2_zipf.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c2_zipf.json
timeout 300 python bench.py --config c3 --no-cpu-baseline --no-pmc --steps 100 > $OUT/bench_c3.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c3.json
timeout 300 python bench.py --config c4 --no-cpu-baseline --steps 100 > $OUT/bench_c4.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c4.json
timeout 300 python bench.py --config c5 --no-cpu-baseline --steps 60 > $OUT/bench_c5_fp16.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c5_fp16.json
timeout 300 python bench.py --no-graph --no-cpu-baseline --no-pmc --no-parity --steps 100 > $OUT/bench_c2_eager_launches.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c2_eager_launches.json
WD_ROW_RECORDS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/bench_c2_uniform_separate_tables.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c2_uniform_separate_tables.json
timeout 300 python bench.py --steps 20 --warmup 5 --force-sharded --no-cpu-baseline --no-pmc --no-parity > $OUT/bench_c2_sharded_one_rank.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_c2_sharded_one_rank.json
scripts/gpu_timeline.sh $TAG/tl --steps 75 > /dev/null 2>&1
cp $OUT/tl/kernel_stats.md $OUT/c2_uniform_kernel_stats.md; cp $OUT/tl/kernel_stats.csv $OUT/c2_uniform_kernel_stats.csv
python - $OUT <<'PY'
import csv, sys
out = sys.argv[1]
ev = []
for r in csv.DictReader(open(out + "/tl/kernel_trace.csv")):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:32]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "")))
ev.sort()
idx = [i for i, e in enumerate(ev) if "tn_group" in e[2]]
a = idx[len(idx) // 2 + 3]
t0 = ev[a][0]
with open(out + "/c2_step_timeline.txt", "w") as f:
    f.write("kernel trace of the bench command (rocprofv3 --kernel-trace), two consecutive steps inside one hipGraph replay;\n"
            "times in us relative to the start of k_gemm_tn_group of the first of them; q = hardware queue\n")
    for e in ev[a - 9: a + 14]:
        f.write("%9.1f %9.1f  %-34s q=%s\n" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, e[2], e[3]))
print(open(out + "/c2_step_timeline.txt").read())
PY
rm -rf $OUT/tl
for rt in 32 16; do WD_CHAIN_RT=$rt python scripts/bench_chain.py 2>&1 | grep -v amdgpu; done > $OUT/tower_chain_stage_cycles.txt; cat $OUT/tower_chain_stage_cycles.txt
python scripts/bench_hgemm.py 2>&1 | grep -v amdgpu > $OUT/hgemm_microbench.txt; python scripts/bench_hgemm_k.py 2>&1 | grep -v amdgpu >> $OUT/hgemm_microbench.txt; cat $OUT/hgemm_microbench.txt
WHICH=nn,tn scripts/gpu_hgemm_pmc.sh $TAG/hg 2>&1 | grep -v amdgpu > $OUT/hgemm_pmc.txt; rm -rf $OUT/hg; head -4 $OUT/hgemm_pmc.txt | cut -c1-300
python scripts/bench_layouts.py 2>&1 | grep -v amdgpu > $OUT/layouts.txt; head -12 $OUT/layouts.txt
scripts/gpu_stats.sh $TAG/c5 40 --config c5 --steps 32 --warmup 8 --pool 8 > /dev/null 2>&1; cp $OUT/c5/kernel_stats.md $OUT/c5_fp16_kernel_stats.md; rm -rf $OUT/c5
