#!/bin/bash
# One rocprofv3 --pmc pass (its own run: counters are not split over passes) of the bench workload.
# usage: bash tools/profile_pmc.sh <tag> "<counters>" [bench args...]
TAG=$1; CNT=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc $CNT --output-format csv -d "$OUT" -o pmc -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 10 --warmup 2 "$@" > "$OUT/bench.log" 2>&1
rc=$?
cd "$REPO"
f=$(find "$OUT" -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python3 - "$f" > "$REPO/gpurun_out/pmc_$TAG.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in rows:
    k = r["Kernel_Name"].split("(")[0]; c = r["Counter_Name"]
    acc[k][c] += float(r["Counter_Value"]); n[k][c] += 1
for k in acc:
    print(k, {c: round(acc[k][c] / n[k][c], 1) for c in acc[k]}, "dispatches", max(n[k].values()))
PY
cat "$REPO/gpurun_out/pmc_$TAG.txt"
rm -rf "$OUT"/*/  2>/dev/null
fi
exit $rc
