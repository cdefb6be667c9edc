"""Summarises the rocprofv3 outputs of tools/collect_profiles.sh: per kernel group and per bench step (1 M reads) the
average duration, HBM bytes (FETCH_SIZE, WRITE_SIZE; KB -> bytes, FETCH_SIZE doubled: gfx950 tallies 128-B requests at
64 B, MI355X_MICROARCH.md) and wave-level VALU / SALU instructions.  usage: pmc_summarize.py <dir> <steps incl. warm-up>"""
import csv
import glob
import json
import os
import re
import sys

GROUPS = [("k_seed", ["k_seed("]), ("k_seed_rows+k_lf_walk+k_seed_final", ["k_seed_rows", "k_lf_walk", "k_seed_final"]),
          ("k_chain", ["k_chain"]), ("k_dp_enum", ["k_dp_enum"]), ("k_ksw_pk", ["k_ksw_pk"]), ("k_ksw (LDS)", ["::k_ksw<"]),
          ("k_stitch+k_finish", ["k_stitch", "k_finish"])]


def group_of(name):
    m = re.search(r"k_ksw_ext<.*?, (\d)>\(", name)
    if m:
        return "k_ksw_ext<%s>" % m.group(1)
    for g, pats in GROUPS:
        if any(p in name for p in pats):
            return g
    return None


def find(d, suffix):
    r = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return r[0] if r else None


def main():
    root, steps = sys.argv[1], int(sys.argv[2])
    out = {}
    st = find(os.path.join(root, "trace"), "kernel_stats.csv")
    if st:
        for row in csv.DictReader(open(st)):
            g = group_of(row["Name"])
            if g:
                o = out.setdefault(g, {})
                o["ms_per_step"] = o.get("ms_per_step", 0.0) + float(row["TotalDurationNs"]) / 1e6 / steps
                o["calls_per_step"] = o.get("calls_per_step", 0.0) + float(row["Calls"]) / steps
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
        f = find(os.path.join(root, "pmc_" + c), "counter_collection.csv")
        if not f:
            continue
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            g = group_of(row["Kernel_Name"])
            if g:
                o = out.setdefault(g, {})
                o[c] = o.get(c, 0.0) + float(row["Counter_Value"]) / steps
    for g, o in out.items():
        if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
            o["hbm_bytes_per_step"] = int(o["FETCH_SIZE"] * 1024 * 2 + o["WRITE_SIZE"] * 1024)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
