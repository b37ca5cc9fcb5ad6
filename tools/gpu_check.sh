#!/bin/bash
# One GPU call: the quick parity tests, then the C3 bench line (no CPU baseline).  Usage: tools/gpu_check.sh <tag> [pytest -k expr]
tag=${1:-x}
sel=${2:-"not full_size"}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q -k "$sel" --timeout 600 --timeout-method thread > gpurun_out/check_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/check_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print(d["frames_per_s"], "fps", d["ms_per_step"], "ms/step", {k: v["ms"] for k, v in d["kernels"].items()}, "gpu_frame", d["gpu_frame_ms"], "timed", d["timed_frames"])
PY
