#!/bin/bash
# Frames/s of every synthetic workload of this repo on one GPU, one line each (not bench lines: see bench.py).
set -e
for w in c1 c2 c3 c5; do echo -n "$w: "; bash tools/bench_kernels.sh --workload $w --steps 100 --warmup 10; done
echo -n "c4: "; bash tools/bench_kernels.sh --workload c4 --steps 30 --warmup 3
echo -n "c3 as rank 3 of 8: "; bash tools/bench_kernels.sh --as-rank 3/8 --steps 200
t() { timeout -k 10 300 python "$@" 2>&1 | grep -v amdgpu.ids | tail -${LINES_:-1} | cut -c1-330; }
t tools/bench_scene.py c2_grid_1m 7680 4320
t tools/bench_scene.py c1_sphere 3840 2160
t tools/bench_walls.py 1
t tools/bench_walls.py 8
t tools/bench_soup.py 3840 2160 5 400 3.0
t tools/bench_soup.py 3840 2160 5 10000 0.3
t tools/bench_soup.py 1920 1080 5 50000 0.15
t tools/bench_sun.py c2_grid_1m
t tools/bench_sun.py c3_grid_10m tex
t tools/bench_settings.py c3_grid_10m 4 1.0 -1
t tools/bench_settings.py c2_grid_1m 1 2.0 -1
t tools/bench_settings.py c3_grid_10m 1 1.0 256
t tools/bench_settings.py c3_grid_10m 1 1.0 0
LINES_=3 t tools/bench_instances.py 10000 quad
