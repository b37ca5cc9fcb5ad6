"""the CPU restatement (oracle/, a port - not the reference) on C1'-C3 with one thread
and with all the threads of the box's CPU share; prints one JSON line per (workload, threads).
Run on the GPU box so the host cores are the ones the GPU numbers sit beside:
    python tools/cpu_table.py > gpurun_out/cpu_table.jsonl
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from renderer_amd import scenes  # noqa: E402


def main():
    many = min(16, os.cpu_count() or 1)
    for name in ("c1", "c2", "c3"):
        builder, desc = bench.WORKLOADS[name]
        scene = getattr(scenes, builder)()
        for threads in (1, many):
            r = bench.cpu_baseline(scene, desc, threads)
            r["workload"] = name
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
