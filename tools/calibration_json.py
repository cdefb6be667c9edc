#!/usr/bin/env python3
"""profiles/rNN_calibration.json out of a tools/calibrate.sh run (gpurun_out/calib_<tag>/valu_mix.txt, gups.txt): the measured
ceilings bench.py quotes its roofline fractions against, tied to the kernel sources of the build they were measured beside.
usage: python tools/calibration_json.py gpurun_out/calib_r05 > profiles/r05_calibration.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = sys.argv[1]
mix = [float(m.group(1)) for m in re.finditer(r"dp_mix\(ksw_ext diagonal\)\s+waves/SIMD=\s*\d+:\s+([0-9.]+) G wave-instr/s", open(os.path.join(d, "valu_mix.txt")).read())]
g64 = [float(m.group(1)) for m in re.finditer(r"bytes/block= 64 dependent=0 waves/CU=\s*\d+:\s+([0-9.]+) G blocks/s", open(os.path.join(d, "gups.txt")).read())]
import bench  # noqa: E402
out = {
    "gather_ceiling_gblocks": max(g64),
    "gather_ceiling_source": "%s/gups.txt: random 64-B blocks (4 x dwordx4 per lane), independent, best of the waves/CU settings" % d,
    "valu_mix_peak_ginst": max(mix),
    "valu_mix_peak_source": "%s/valu_mix.txt: dp_mix (the opcode mix of one ksw_ext diagonal, tools/valu_mix.hip), best of 1..8 waves/SIMD, all 1024 SIMDs" % d,
    "kernel_source_hash": bench.kernel_source_hash(),
}
print(json.dumps(out, indent=1))
