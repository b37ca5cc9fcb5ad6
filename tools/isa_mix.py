"""Static instruction mix of one kernel of a gfx950 assembly listing, per basic block.

    hipcc ... -S --cuda-device-only kernels.hip -o kernels.s
    python tools/isa_mix.py kernels.s '<mangled-name substring>' [--blocks]

Classes follow tools/microbench/valu_ops (profiles/r02_valu_ops_microbench.txt): "fast" = f32 fma/mul/add/sub, v_mov,
v_add_u32, v_and/or/xor with VGPR or inline-constant operands (2.25 SIMD cycles per wave64 instruction); "trans" =
rcp/rsq/sqrt/exp/log/sin/cos (8.2); everything else VALU, and any VALU instruction with an SGPR or literal operand, "slow" (4.2).
"""
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_mac_f32", "v_mad_f32"}
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def classify(line):
    op = line.split()[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if not base.startswith("v_"):
        if base.startswith("s_"):
            return "salu"
        if base.startswith(("global_", "flat_", "buffer_", "scratch_")):
            return "vmem"
        if base.startswith("ds_"):
            return "lds"
        return "other"
    if base.startswith(TRANS):
        return "trans"
    args = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
    srcs = args.split(",")[1:]
    sgpr = any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|0x[0-9a-f]{3,})", a) for a in srcs)
    if base in FAST and not sgpr and "clamp" not in op:
        return "fast"
    return "slow"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and l.rstrip().endswith(l.split(":")[0].strip() and l.rstrip()[-1]))
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith("s_endpgm"):
            cur[1].append(s)
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
            continue
        if not s or s.startswith((";", ".")):
            continue
        cur[1].append(s.split(";")[0].strip())
    blocks.append(cur)
    total = {}
    for name, ins in blocks:
        c = {}
        for i in ins:
            k = classify(i)
            c[k] = c.get(k, 0) + 1
            total[k] = total.get(k, 0) + 1
        if show_blocks and ins:
            tgt = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
            print(f"{name:14s} n={len(ins):4d} {c} -> {tgt}")
    print("total", total)


if __name__ == "__main__":
    main()
