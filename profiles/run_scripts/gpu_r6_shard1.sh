cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6shard1
MASTER_PORT=29571 timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --force-sharded --no-cpu-baseline --no-pmc > gpurun_out/r6shard1/bench_c4_sharded_one_rank.json 2> gpurun_out/r6shard1/c4.err; tail -n 5 gpurun_out/r6shard1/c4.err
MASTER_PORT=29572 timeout 600 python bench.py --steps 20 --warmup 5 --force-sharded --no-cpu-baseline --no-pmc > gpurun_out/r6shard1/bench_c2_sharded_one_rank.json 2> gpurun_out/r6shard1/c2.err; tail -n 3 gpurun_out/r6shard1/c2.err
python - <<'PY'
import json
for f in ("bench_c4_sharded_one_rank","bench_c2_sharded_one_rank"):
    try:
        d=json.loads(open("gpurun_out/r6shard1/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d.get("repeats_ms_per_step"), d.get("parity"), d["config"].get("exchange",{}).get("graph"))
    except Exception as e: print(f,"FAILED",e)
PY
