"""Registers, spills and LDS of every kernel of an assembly listing whose name contains the given substring."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+)\n.*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", s, re.S):
    if pat in m.group(1):
        lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", m.group(0))
        print(m.group(1)[:60], "vgpr", m.group(3), "spill", m.group(4), "sgpr", m.group(2), "lds", lds.group(1) if lds else "?")
