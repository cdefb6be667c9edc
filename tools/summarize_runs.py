#!/usr/bin/env python3
"""One line per bench detail file (bench.py --detail-file): leg, batches in flight, reads/s, ms per step, per-stage kernel times,
seeding fraction of the gather ceiling, parity.  The experiment tables under profiles/ (r06_seed_task_leaf.txt, r06_smem_tasks_tuning.txt,
r06_h2h_inflight.txt ...) are this script's output over the detail files of one gpurun call.
usage: summarize_runs.py <glob of detail files> [<glob> ...]"""
import glob
import json
import os
import sys

for pat in sys.argv[1:]:
    for f in sorted(glob.glob(pat)):
        tag = os.path.basename(f).replace(".json", "")
        try:
            d = json.load(open(f))
        except ValueError:
            print("%-34s (no result)" % tag)
            continue
        for w in d["workloads"]:
            legs = [("", w)] + [(k, w[k]) for k in ("overlapped", "host_to_host") if isinstance(w.get(k), dict) and "value" in w[k]]
            for name, l in legs:
                rf = l.get("roofline") or {}
                km = l.get("kernel_ms_per_step_under_overlap") or rf.get("kernel_ms_per_step") or {}
                pc = (l.get("cpu_baseline") or {}).get("parity_check") or {}
                xp = l.get("cross_leg_parity") or {}
                print("%-34s %-12s %s in flight, %-14s %11.1f reads/s %9.1f ms/step  %s%s%s%s" % (
                    tag, name or w["name"], l.get("batches_in_flight", 1), (l.get("io") or "")[:14], l["value"], l["ms_per_step"],
                    " ".join("%s %.1f" % (k.split("+")[0], v) for k, v in km.items()),
                    "  seeding %.3f of the gather ceiling" % rf["seeding_frac_of_gather_ceiling"] if "seeding_frac_of_gather_ceiling" in rf else "",
                    "  parity vs oracle: %d mismatching of %d reads" % (pc["mismatching_reads"], pc["reads"]) if pc else "",
                    "  vs the same steps alone: %d mismatching of %d steps" % (xp["mismatching_steps"], xp["steps_compared"]) if xp else ""))
                for key in ("narrow_band", "narrow_band_long"):
                    if rf.get(key):
                        print("%-34s   %s: %s" % ("", key, json.dumps(rf[key])))
