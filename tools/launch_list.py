#!/usr/bin/env python3
"""Per-launch durations of selected kernels from a rocprofv3 --kernel-trace csv (diagnostics).
usage: launch_list.py <dir with *kernel_trace.csv> [substring ...]"""
import csv, glob, re, sys
d = sys.argv[1]
pats = sys.argv[2:] or ["k_ksw", "k_chain", "k_stitch", "k_seed"]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows:
    n = r["Kernel_Name"]
    if not any(p in n for p in pats):
        continue
    m = re.search(r"(k_\w+)(<[^>]*?, (\d)>)?", n.replace("(anonymous namespace)::", ""))
    short = m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")
    print("%-22s grid %7s  start %10.3f  %9.3f ms" % (short, r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["Start_Timestamp"]) - t0) / 1e6,
                                                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
