"""Summarises the rocprofv3 outputs of tools/collect_profiles.sh: per kernel group and per bench step the average
duration, HBM bytes (FETCH_SIZE, WRITE_SIZE in KB) and wave-level VALU / SALU instructions.
FETCH_SIZE counts fabric read requests x 64 B; a request is 64 B for the random occ-block gathers of k_seed / k_lf_walk
(raw value = bytes, calibrated on tools/gups launches of known size: profiles/r02_counter_calibration.txt) but 128 B for
wide coalesced streams (the x2 of MI355X_MICROARCH.md).  Kernels with a mixed pattern get both readings (hbm_bytes_min /
_max) and the x2 reading as hbm_bytes_per_step.
usage: pmc_summarize.py <dir> <steps incl. warm-up> [<out.json> <read_len> <reads_per_step> <preset>]"""
import csv
import glob
import json
import os
import re
import sys

# every kernel of the pipeline's stages belongs to exactly one group; the DP families each have their own
GROUPS = [("k_seed", ["k_seed(", "k_seed<", "k_seed_long(", "k_seed_long<", "k_seed_tasks", "k_task_", "k_mems"]),
          ("k_seed_rows+k_lf_walk+k_seed_final", ["k_seed_rows", "k_lf_walk", "k_seed_final", "k_seg_seed_counts", "k_read_seed_ranges"]),
          ("k_chain", ["k_chain", "k_soc_windows", "k_soc_dump", "k_sort_seeds_wave", "k_hseed_counts", "k_hset_flatten"]),
          ("k_dp_enum", ["k_dp_enum", "k_job_cost", "k_grp_hist", "k_grp_scatter", "k_ops_caps"]),
          ("k_ksw_band", ["k_ksw_band<"]), ("k_ksw_pk", ["k_ksw_pk"]), ("k_ksw (LDS)", ["::k_ksw<"]),
          ("k_stitch+k_finish", ["k_stitch", "k_finish"])]
# a kernel whose name carries one of these and that matches no group is an ERROR (round 5: k_ksw_band fell through silently and the
# DP roofline counted four of five kernel families)
MUST_MATCH = ("k_ksw", "k_seed", "k_chain", "k_stitch", "k_soc", "k_dp_enum", "k_lf_walk")
UNMATCHED = set()


def group_of(name):
    m = re.search(r"k_ksw_ext<.*?, (\d)>\(", name)
    if m:
        return "k_ksw_ext<%s>" % m.group(1)
    m = re.search(r"k_ksw_band<.*?, (true|false), (\d)>\(", name)
    if m:
        return "k_ksw_band" if m.group(2) == "4" else "k_ksw_band (long)"  # four short jobs per wave / one long job per wave (ksw_band.h)
    m = re.search(r"k_ksw_grp<.*?, (\d), (\d), (true|false)>\(", name)
    if m:
        return "k_ksw_grp<%s>" % m.group(1)  # (several short extension jobs per wavefront: ksw_grp.h)
    for g, pats in GROUPS:
        if any(p in name for p in pats):
            return g
    if any(k in name for k in MUST_MATCH):
        UNMATCHED.add(name)
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
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    return {"workload_key": workload_key,
            "kernel_source_hash": kernel_source_hash(),  # bench.py replays these numbers only for this build of the kernels
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / SQ_INSTS_SALU in separate passes of `bench.py --workload .. "
                    "--steps 3 --warmup 1 --cpu-sample 0` (tools/collect_profiles.sh); FETCH_SIZE (KB) = fabric read requests x 64 B: "
                    "taken as bytes for the random 64-B gathers (k_seed, k_lf_walk; calibrated in profiles/r02_counter_calibration.txt) "
                    "and doubled for the other kernels (128-B requests, MI355X_MICROARCH.md); per step = per launch of each stage",
            "bytes_per_launch": b,
            "valu_wave_insts_per_launch": {"k_ksw": tot("SQ_INSTS_VALU")},
            "salu_insts_per_launch": {"k_ksw": tot("SQ_INSTS_SALU")},
            "per_kernel": summary,
            "note_valu": "SQ_INSTS_VALU = wave-level VALU instructions per step; measured issue rate of the DP kernels' opcode mix: "
                         "576.9 G wave-inst/s over 1024 SIMDs (profiles/r02_valu_mix.txt); the scalar unit of a CU issues one "
                         "instruction per cycle"}


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
    GATHER = ("k_seed", "k_seed_rows+k_lf_walk+k_seed_final")  # random 64-B blocks: FETCH_SIZE needs no doubling
    for g, o in out.items():
        if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
            lo = int(o["FETCH_SIZE"] * 1024 + o["WRITE_SIZE"] * 1024)
            hi = int(o["FETCH_SIZE"] * 1024 * 2 + o["WRITE_SIZE"] * 1024)
            o["hbm_bytes_min"], o["hbm_bytes_max"] = lo, hi
            o["fetch_request_bytes"] = 64 if g in GATHER else 128
            o["hbm_bytes_per_step"] = lo if g in GATHER else hi
    if UNMATCHED:
        sys.stderr.write("pmc_summarize: kernels of the pipeline that match no group (add them to GROUPS):\n  " + "\n  ".join(sorted(UNMATCHED)) + "\n")
        sys.exit(2)
    # what bench.py's top-level roofline replays: the sum over EVERY k_ksw* row
    ksw = [g for g in out if g.startswith("k_ksw")]
    totals = {"rows": sorted(ksw)}
    for key in ("ms_per_step", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "hbm_bytes_per_step"):
        if any(key in out[g] for g in ksw):
            totals[key] = sum(out[g].get(key, 0) for g in ksw)
    shown = dict(out)
    shown["_sum_over_all_k_ksw_rows"] = totals
    print(json.dumps(shown, indent=1))
    if len(sys.argv) > 3:
        key = [int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], 1.0] if len(sys.argv) > 6 else [150, 1000000, "default", 1.0]
        entry = traffic_json(out, key)
        doc = {"workloads": []}
        if os.path.exists(sys.argv[3]):
            try:
                doc = json.load(open(sys.argv[3]))
            except ValueError:
                pass
        doc["workloads"] = [w for w in doc.get("workloads", []) if w.get("workload_key") != key] + [entry]
        with open(sys.argv[3], "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()

