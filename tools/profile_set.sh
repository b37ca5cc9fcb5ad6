#!/bin/bash
# The whole profile set of one kernel state on the GPU box (kernel stats serial and overlapped, four PMC passes, the
# bench line with its CPU baseline, the workload table).  usage: bash tools/profile_set.sh <tag>   (e.g. r02_v8)
# Outputs land in gpurun_out/; copy the summaries into profiles/ under the tag (profiles/README.md) and run
# tools/make_traffic.py on the three PMC summaries.
T=${1:-set}
set -o pipefail
RND_OVERLAP=0 bash tools/profile_stats.sh ${T}s --steps 200 --warmup 20 > /dev/null || exit 1
bash tools/profile_stats.sh ${T}o --steps 200 --warmup 20 > /dev/null || exit 1
bash tools/profile_pmc.sh ${T}_fetch "FETCH_SIZE" > /dev/null || exit 1
bash tools/profile_pmc.sh ${T}_write "WRITE_SIZE" > /dev/null || exit 1
bash tools/profile_pmc.sh ${T}_sq "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" > /dev/null || exit 1
bash tools/profile_pmc.sh ${T}_ta "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE" > /dev/null || exit 1
timeout -k 10 400 python bench.py --steps 200 --warmup 20 2> gpurun_out/bench_${T}.err | tail -1 > gpurun_out/bench_${T}.json || exit 1
cut -c1-200 gpurun_out/bench_${T}.json
bash tools/workload_table.sh > gpurun_out/workloads_${T}.txt 2>&1 || exit 1
tail -3 gpurun_out/workloads_${T}.txt | cut -c1-200
