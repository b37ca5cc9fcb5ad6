set -o pipefail
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gpu_all3.log 2>&1 || { tail -20 gpurun_out/r2_gpu_all3.log; exit 1; }
tail -2 gpurun_out/r2_gpu_all3.log
RND_STORE_CLIP=0 timeout -k 10 400 python -u tools/fuzz_parity.py 200 7000 > gpurun_out/fuzz_rederive.txt 2>&1 || { tail -5 gpurun_out/fuzz_rederive.txt; exit 1; }
tail -2 gpurun_out/fuzz_rederive.txt
RND_STORE_CLIP=1 timeout -k 10 400 python -u tools/fuzz_parity.py 200 7200 > gpurun_out/fuzz_stored.txt 2>&1 || { tail -5 gpurun_out/fuzz_stored.txt; exit 1; }
tail -2 gpurun_out/fuzz_stored.txt
echo "gloo 2 ranks sort-first c3"
RND_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/gloo_c3.err | tail -1 | cut -c1-400 || { tail -20 gpurun_out/gloo_c3.err; exit 1; }
echo "gloo 2 ranks sort-last c5"
RND_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --workload c5 2> gpurun_out/gloo_c5.err | tail -1 | cut -c1-400 || { tail -20 gpurun_out/gloo_c5.err; exit 1; }
