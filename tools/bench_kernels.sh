#!/bin/bash
# prints fps + per-kernel ms of bench.py for the given extra args; env vars pass through
timeout -k 10 400 python -u bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['frames_per_s'], d['ms_per_step'], d['gpu_frame_ms'], {k:v['ms'] for k,v in d['kernels'].items()})"
