#!/bin/bash
# Two ranks on the one GPU of the box, gloo for the host-side collectives: bench.py's N>1 paths with the frame check.
set -o pipefail
echo "gloo 2 ranks sort-first c3"
RND_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 2> gpurun_out/gloo_c3.err | tail -1 > gpurun_out/gloo_c3.json || { tail -20 gpurun_out/gloo_c3.err; exit 1; }
python3 -c "import json; d=json.load(open('gpurun_out/gloo_c3.json')); print(d['frames_per_s'], d['frame_check'], d['config']['partition'])" || exit 1
echo "gloo 2 ranks sort-last c5"
RND_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 5 --workload c5 2> gpurun_out/gloo_c5.err | tail -1 > gpurun_out/gloo_c5.json || { tail -20 gpurun_out/gloo_c5.err; exit 1; }
python3 -c "import json; d=json.load(open('gpurun_out/gloo_c5.json')); print(d['frames_per_s'], d['frame_check'], d['config']['partition'])" || exit 1
