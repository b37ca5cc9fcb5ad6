"""Per-frame GPU timeline from a rocprofv3 trace directory (kernel_trace.csv and, when present,
memory_copy_trace.csv): busy time, idle gaps between consecutive operations, and the mean duration
and preceding gap of each operation of a steady-state frame.

    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o tl --output-format csv -- python3 bench.py ...
    python tools/timeline.py gpurun_out/tl
"""
import csv, glob, os, sys
from collections import defaultdict


def load(root):
    ops = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("rnd::", "").replace("void ", "")
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    for path in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?").replace("MEMORY_COPY_", "")))
    ops.sort()
    return ops


def main():
    ops = load(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else "k_clear"
    # frames start at the first kernel of the sequence; use the last 60 % of them
    starts = [i for i, o in enumerate(ops) if o[2].startswith(first)]
    if len(starts) < 8:
        print("too few frames", len(starts)); return
    lo = starts[len(starts) * 4 // 10]
    hi = starts[-2]
    sel = ops[lo:hi]
    n_frames = sum(1 for i in starts if lo <= i < hi)
    dur = defaultdict(float); gap = defaultdict(float); cnt = defaultdict(int)
    prev_end = sel[0][0]
    order = []
    for s, e, n in sel:
        if n not in cnt: order.append(n)
        dur[n] += e - s; gap[n] += max(0, s - prev_end); cnt[n] += 1
        prev_end = max(prev_end, e)
    span = sel[-1][1] - sel[0][0]
    print("frames %d  span/frame %.1f us  busy/frame %.1f us  idle/frame %.1f us" % (
        n_frames, span / n_frames / 1e3, sum(dur.values()) / n_frames / 1e3, sum(gap.values()) / n_frames / 1e3))
    print("%-28s %8s %10s %10s" % ("operation", "per frm", "dur us", "gap-before us"))
    for n in order:
        print("%-28s %8.2f %10.2f %10.2f" % (n[:28], cnt[n] / n_frames, dur[n] / cnt[n] / 1e3, gap[n] / cnt[n] / 1e3))


if __name__ == "__main__":
    main()
