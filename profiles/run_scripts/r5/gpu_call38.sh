#!/bin/bash
# round 5 call 38: lines moved by configs[3]'s memory-bound kernels (two --pmc passes, request counters by size)
O=gpurun_out/r5_call38; mkdir -p $O
export TMPDIR=/tmp
B="--config c4 --steps 8 --warmup 4 --pool 4 --repeats 1 --no-cpu-baseline --no-pmc --no-parity --no-graph"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/rd -o pmc -- python bench.py $B > $O/rd.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/wr -o pmc -- python bench.py $B > $O/wr.log 2>&1
python scripts/pmc_kernel_traffic.py $O/rd $O/wr k_bucket_update k_embag_fwd_range k_wide_fwd k_small_fwd k_small_bwd k_bucket_hist k_bucket_scatter k_tower_chain8 k_gemm_tn_group > $O/c4_kernel_traffic.txt 2>&1; cat $O/c4_kernel_traffic.txt
rm -rf $O/rd $O/wr
