"""profiles/traffic.json from the summaries of the rocprofv3 --pmc passes (tools/profile_pmc.sh).

usage: python tools/make_traffic.py <fetch.txt> <write.txt> <sq.txt> <profile tag> [workload]   (-> profiles/traffic_<workload>.json)
HBM bytes per launch = 2 * FETCH_SIZE * 1000 + WRITE_SIZE * 1000 (FETCH_SIZE and WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reads half of the bytes fetched: MI355X_MICROARCH.md, HBM section; checked on k_transform, whose
12 B/vertex of reads are 60.2 MB against FETCH_SIZE 31.3 MB).
"""
import ast, hashlib, json, os, sys

KEYS = {"k_clear": "clear", "k_cull": "cull", "k_transform": "transform", "k_setup": "setup", "k_hiz": "setup", "k_cull_far": "setup", "k_transform<8": "setup", "k_clip": "clip",
        "k_scan_tiles": "scan", "k_bin_fill": "bin", "k_raster": "raster", "k_shade": "shade"}


def read(path):
    """Per FRAME: every dispatch of every variant of a kernel (the two passes of the early-depth k_setup and the k_hiz
    between them included, as bench.py times them: events around the whole launch_setup), summed over the run and
    divided by the run's frames (= dispatches of k_shade).  The first frames of a C4 run, before the host has switched
    the early-depth variant on, are part of that mean (2 to 4 of ~216)."""
    total, frames = {}, 0
    for line in open(path):
        name, _, rest = line.partition(" {")
        body, _, tail = rest.rpartition("}")
        n = int(tail.split()[-1])
        vals = ast.literal_eval("{" + body + "}")
        for k, short in KEYS.items():
            if k == "k_transform" and "k_transform<8" in name:
                continue        # (the second launch of an early-depth frame: timed with the setup it belongs to)
            if "rnd::" + k + "<" in name + "<" or ("rnd::" + k) == name.strip() or (k.endswith("<8") and "rnd::" + k in name):
                acc = total.setdefault(short, {})
                for c, v in vals.items():
                    acc[c] = acc.get(c, 0.0) + v * n
                if k == "k_shade":
                    frames += n
    return {k: {c: v / max(frames, 1) for c, v in acc.items()} for k, acc in total.items()}


fetch, write, sq, tag = read(sys.argv[1]), read(sys.argv[2]), read(sys.argv[3]), sys.argv[4]
workload = sys.argv[5] if len(sys.argv) > 5 else "c3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(open(os.path.join(ROOT, "renderer_amd", "csrc", "kernels.hip"), "rb").read()).hexdigest()
kernels = {k: int(round(2000.0 * fetch[k]["FETCH_SIZE"] + 1000.0 * write[k]["WRITE_SIZE"])) for k in fetch if k in write}
valu = {k: int(v["SQ_INSTS_VALU"]) for k, v in sq.items() if v.get("SQ_INSTS_VALU", 0) > 0}
json.dump({
    "workload": workload, "n_gpus": 1,
    "kernels_sha256": sha,      # of renderer_amd/csrc/kernels.hip when the passes ran: bench.py drops the counters when the tree's differs
    "source": f"rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc SQ_INSTS_VALU ... in separate runs (tools/profile_pmc.sh; "
              f"summaries profiles/{tag}_pmc_fetch_{workload}.txt, {tag}_pmc_write_{workload}.txt, {tag}_pmc_sq_{workload}.txt), mean per frame over the frames of bench.py --workload {workload}",
    "correction": "gfx950: FETCH_SIZE (KB) reads 1/2 of the bytes fetched (MI355X_MICROARCH.md, HBM): bytes = 2*FETCH_SIZE*1000 + WRITE_SIZE*1000",
    "grouping": "per frame, one entry per kernel as bench.py times them (k_clear, k_cull, k_transform, k_setup incl. its two early-depth passes and k_hiz, k_clip, k_scan_tiles, k_bin_fill, k_raster, k_shade); all passes of a frame summed",
    "kernels": kernels, "valu_insts": valu}, open(os.path.join(ROOT, "profiles", f"traffic_{workload}.json"), "w"), indent=2)
print(json.dumps({"kernels": kernels, "valu_insts": valu}, indent=1))
