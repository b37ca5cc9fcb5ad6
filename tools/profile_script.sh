#!/bin/bash
# rocprofv3 --kernel-trace --stats of any python script of this repo.  usage: bash tools/profile_script.sh <tag> <script.py> [args...]
set -e
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
S=$REPO/$1; shift
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 "$S" "$@" > "$OUT/run.log" 2>&1
cd "$REPO"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$REPO/gpurun_out/kernel_stats_$TAG.csv"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
cut -c1-160 "$REPO/gpurun_out/kernel_stats_$TAG.csv"
