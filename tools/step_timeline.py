#!/usr/bin/env python3
"""Every launch of ONE step of a single-stream bench run from a rocprofv3 --kernel-trace csv: name, start, duration and the gap
to the previous launch's end -- where a step's time goes besides its big kernels (the small scan / sort / fill
launches).  A step starts at a k_seed / k_seed_long / k_seed_tasks / k_mems launch; the LAST complete step is printed.
usage: step_timeline.py <dir with *kernel_trace.csv> [--all]"""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("ma::", "")
    m = re.search(r"rocprim::.*?detail::(\w+)", n)
    if m:
        k = re.search(r"(radix_sort\w*|scan\w*|lookback\w*|transform\w*|reduce\w*|partition\w*|histogram\w*|merge\w*)", n)
        return "rocprim:" + (k.group(1) if k else m.group(1))[:40]
    m = re.match(r"(?:void )?(\w+)(<.*)?", n)
    if not m:
        return n[:40]
    t = re.search(r", (\d)>", n)
    return m.group(1) + ("<%s>" % t.group(1) if t and m.group(1).startswith("k_ksw") else "")


starts = [i for i, r in enumerate(rows) if re.search(r"\bk_(seed|seed_long|seed_tasks|mems)\b|k_seed<", r["Kernel_Name"])]
if len(starts) < 2:
    sys.exit("fewer than two steps in the trace")
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
tot_k = tot_gap = 0.0
small = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    print("%-44s start %9.1f us  dur %9.1f us  gap %8.1f us" % (short(r["Kernel_Name"]), (s - t0) / 1e3, dur, gap))
    tot_k += dur
    tot_gap += max(gap, 0.0)
    prev_end = max(prev_end, e)
    if dur < 500:
        small.append(dur)
wall = (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3
print("step: %d launches, wall %.1f us, kernels %.1f us, gaps %.1f us (last launch to next step %.1f us); %d launches under 0.5 ms: %.1f us"
      % (len(step), wall, tot_k, tot_gap, (int(rows[hi]["Start_Timestamp"]) - prev_end) / 1e3, len(small), sum(small)))
