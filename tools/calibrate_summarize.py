"""Summarises tools/calibrate.sh: raw rocprofv3 counters of tools/gups launches against their known byte counts."""
import csv, glob, os, re, sys

root = sys.argv[1]
rows = []
for log in sorted(glob.glob(os.path.join(root, "pmc_*.log"))):
    m = re.match(r"pmc_(.*)_(FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum|TCC_EA0_RDREQ_32B_sum)\.log", os.path.basename(log))
    if not m:
        continue
    cfg, ctr = m.group(1), m.group(2)
    known = None
    for line in open(log, errors="replace"):
        k = re.search(r"known_bytes=(\d+)", line)
        if k:
            known = int(k.group(1))
    f = glob.glob(os.path.join(root, "pmc_%s_%s" % (cfg, ctr), "**", "*counter_collection.csv"), recursive=True)
    val = 0.0
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") == ctr and ("k_gather" in r["Kernel_Name"] or "k_stream" in r["Kernel_Name"]):
                val = max(val, float(r["Counter_Value"]))  # the timed launch is the largest one
    rows.append((cfg, ctr, known, val))
print("%-18s %-24s %16s %16s %s" % ("launch", "counter", "known bytes", "raw value", "known / (raw x unit)"))
for cfg, ctr, known, val in rows:
    unit = 1024.0 if ctr in ("FETCH_SIZE", "WRITE_SIZE") else (32.0 if "32B" in ctr else 64.0)
    ratio = (known / (val * unit)) if (known and val) else float("nan")
    print("%-18s %-24s %16s %16.0f %.3f" % (cfg, ctr, known, val, ratio))
