#!/bin/bash
# rocprofv3 kernel trace of a bench run + the per-frame timeline (tools/timeline.py).  usage: bash tools/timeline.sh <tag> [bench args...]
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/tl_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o tl -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
cd "$REPO"
python3 tools/timeline.py "$OUT" | tee gpurun_out/timeline_$TAG.txt
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
