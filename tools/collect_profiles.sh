#!/bin/bash
# Copies the summaries tools/profile_set.sh left in gpurun_out/ (merged back from the GPU box) into profiles/ (tracked).
#   usage: bash tools/collect_profiles.sh <tag> <workload>
T=$1; W=$2
cp gpurun_out/kernel_stats_${T}s_$W.csv profiles/${T}_kernel_stats_$W.csv
cp gpurun_out/bench_under_rocprof_${T}s_$W.json profiles/${T}_bench_under_rocprof_$W.json
[ -f gpurun_out/kernel_stats_${T}o_$W.csv ] && cp gpurun_out/kernel_stats_${T}o_$W.csv profiles/${T}_kernel_stats_${W}_overlapped.csv
for k in fetch write sq; do cp gpurun_out/pmc_${T}_${k}_$W.txt profiles/${T}_pmc_${k}_$W.txt; done
[ -f gpurun_out/pmc_${T}_ta_$W.txt ] && cp gpurun_out/pmc_${T}_ta_$W.txt profiles/${T}_pmc_ta_tcp_$W.txt
cp gpurun_out/traffic_$W.json profiles/traffic_$W.json
cp gpurun_out/bench_${T}_$W.json profiles/${T}_bench_$W.json
ls profiles/${T}_*_$W* profiles/traffic_$W.json
