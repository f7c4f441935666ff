#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in an AMDGPU assembly listing
(hipcc -S --cuda-device-only): binary64 VALU, other VALU, compares, SALU, vector
memory, LDS, scratch, and where the block branches to.
usage: isa_blocks.py listing.s mangled-name-substring"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", collections.Counter()]
blocks.append(cur)
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), collections.Counter()]
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    c = cur[1]
    if op.startswith("v_cmp"):
        c["vcmp"] += 1
    elif op.startswith("v_"):
        c["vf64" if "f64" in op else "v32"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        c["tgt:" + t.split()[-1]] += 0
    elif op.startswith("s_"):
        c["s"] += 1
    elif op.split("_")[0] in ("global", "buffer", "flat"):
        c["vmem"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith("scratch_"):
        c["scr"] += 1
for name, c in blocks:
    tg = ",".join(k[4:] for k in c if k.startswith("tgt:"))
    print("%-12s vf64 %3d v32 %3d vcmp %2d s %3d vmem %2d lds %2d scr %d -> %s" % (
        name, c["vf64"], c["v32"], c["vcmp"], c["s"], c["vmem"], c["lds"], c["scr"], tg))
