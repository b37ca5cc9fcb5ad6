#!/bin/bash
# Per-kernel time of the bench workload: rocprofv3 --kernel-trace --stats.
# usage (on the GPU box, from the repo root): bash tools/profile_stats.sh <tag> [bench args...]
set -e
TAG=${1:-run}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
cd "$REPO"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$REPO/gpurun_out/kernel_stats_$TAG.csv"
tail -1 "$OUT/bench.log" > "$REPO/gpurun_out/bench_under_rocprof_$TAG.json"
# keep the merge small: the per-dispatch trace can be large
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
cat "$REPO/gpurun_out/kernel_stats_$TAG.csv"
