#!/usr/bin/env python3
"""Registers, spills, scratch and static LDS of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only): the code
object's metadata, one line per kernel whose name contains the pattern.   usage: kernel_regs.py file.s [pattern]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk.split("\n  - ")[0]))
    if pat in f.get("name", ""):
        print("%-100s vgpr %3s spill %3s  sgpr %3s spill %3s  scratch %4s  lds %6s" % (
            f["name"][:100], f.get("vgpr_count"), f.get("vgpr_spill_count"), f.get("sgpr_count"), f.get("sgpr_spill_count"),
            f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
