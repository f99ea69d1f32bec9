#!/bin/bash
# round 6: lines moved by configs[3]'s memory-bound kernels on the final tree (two --pmc passes, request counters by size; eager launches:
# counter passes serialise kernels) -- profiles/r6_c4_kernel_traffic.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${1:-r6c4traffic}; mkdir -p $O
B="--config c4 --steps 8 --warmup 4 --pool 4 --repeats 1 --no-cpu-baseline --no-pmc --no-parity --no-graph"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/rd -o pmc -- python bench.py $B > $O/rd.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/wr -o pmc -- python bench.py $B > $O/wr.log 2>&1
python scripts/pmc_kernel_traffic.py $O/rd $O/wr k_row_update "k_embag_fwd<4, true>" k_wide_sum k_small_fwd k_small_bwd k_bucket_hist k_bucket_colscan k_bucket_scatter k_bucket_sort_wave k_feat_emit_par k_fingerprint64 k_tower_chain8 k_gemm_tn_group > $O/c4_kernel_traffic.txt 2>&1; cat $O/c4_kernel_traffic.txt
rm -rf $O/rd $O/wr
