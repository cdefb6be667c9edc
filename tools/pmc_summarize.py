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



def traffic_json(summary, workload_key):
    """profiles/r01_pmc_traffic.json in the schema bench.py reads (k_ksw = all DP kernels of a step)."""
    ksw = [g for g in summary if g.startswith("k_ksw")]
    tot = lambda key: sum(summary[g].get(key, 0) for g in ksw)
    b = {g: o["hbm_bytes_per_step"] for g, o in summary.items() if not g.startswith("k_ksw") and "hbm_bytes_per_step" in o}
    b["k_ksw"] = int(tot("hbm_bytes_per_step"))
    return {"workload_key": workload_key,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / SQ_INSTS_SALU in separate passes of `bench.py --steps 3 "
                    "--warmup 1 --cpu-sample 0` (tools/collect_profiles.sh); KB -> bytes, FETCH_SIZE doubled (MI355X_MICROARCH.md: "
                    "gfx950 tallies 128-B requests at 64 B); per step = per launch of each stage",
            "bytes_per_launch": b,
            "valu_wave_insts_per_launch": {"k_ksw": tot("SQ_INSTS_VALU")},
            "salu_insts_per_launch": {"k_ksw": tot("SQ_INSTS_SALU")},
            "per_kernel": summary,
            "note_valu": "SQ_INSTS_VALU = wave-level VALU instructions per step; a wave64 VALU instruction occupies its SIMD for 4 "
                         "cycles, 1024 SIMDs at 2.4 GHz = 614.4 G wave-inst/s; the scalar unit of a CU issues one instruction per cycle"}


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
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(traffic_json(out, [150, 1000000, "default", 1.0]), f, indent=1)


if __name__ == "__main__":
    main()

