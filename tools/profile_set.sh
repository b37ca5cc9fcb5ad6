#!/bin/bash
# The profile set of one workload on the GPU box: rocprofv3 kernel stats (serial, and overlapped for c3), the FETCH /
# WRITE / SQ (/ TA for c3) PMC passes, each its own run, and the traffic file bench.py reads.
#   usage: bash tools/profile_set.sh <tag> <workload> [steps]      (e.g. r03_v1 c4 30)
# Outputs land in gpurun_out/ AND (summaries only) in profiles/<tag>_*_<workload>.*; profiles/traffic_<workload>.json is rebuilt.
T=${1:-set}; W=${2:-c3}; STEPS=${3:-200}
WARM=$(( STEPS / 10 + 2 ))
set -o pipefail
A="--workload $W"
RND_OVERLAP=0 bash tools/profile_stats.sh ${T}s_$W $A --steps $STEPS --warmup $WARM > /dev/null || exit 1
cp gpurun_out/kernel_stats_${T}s_$W.csv profiles/${T}_kernel_stats_$W.csv
cp gpurun_out/bench_under_rocprof_${T}s_$W.json profiles/${T}_bench_under_rocprof_$W.json
if [ "$W" = c3 ]; then
  bash tools/profile_stats.sh ${T}o_$W $A --steps $STEPS --warmup $WARM > /dev/null || exit 1
  cp gpurun_out/kernel_stats_${T}o_$W.csv profiles/${T}_kernel_stats_${W}_overlapped.csv
fi
bash tools/profile_pmc.sh ${T}_fetch_$W "FETCH_SIZE" $A > /dev/null || exit 1
bash tools/profile_pmc.sh ${T}_write_$W "WRITE_SIZE" $A > /dev/null || exit 1
bash tools/profile_pmc.sh ${T}_sq_$W "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" $A > /dev/null || exit 1
for k in fetch write sq; do cp gpurun_out/pmc_${T}_${k}_$W.txt profiles/${T}_pmc_${k}_$W.txt; done
if [ "$W" = c3 ]; then
  bash tools/profile_pmc.sh ${T}_ta_$W "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE" $A > /dev/null || exit 1
  cp gpurun_out/pmc_${T}_ta_$W.txt profiles/${T}_pmc_ta_tcp_$W.txt
fi
python3 tools/make_traffic.py profiles/${T}_pmc_fetch_$W.txt profiles/${T}_pmc_write_$W.txt profiles/${T}_pmc_sq_$W.txt $T $W > /dev/null || exit 1
cp profiles/traffic_$W.json gpurun_out/traffic_$W.json
timeout -k 10 400 python bench.py $A --steps $STEPS --warmup $WARM 2> gpurun_out/bench_${T}_$W.err | tail -1 > gpurun_out/bench_${T}_$W.json || exit 1
cp gpurun_out/bench_${T}_$W.json profiles/${T}_bench_$W.json
cut -c1-300 gpurun_out/bench_${T}_$W.json
