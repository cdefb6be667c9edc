#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run: per kernel group the summed duration, and how much of the wall time
had 1, 2, 3.. kernels running at once.  usage: timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_" not in n and "rocprim" not in n and "hipcub" not in n:
        continue
    # "void ma::k_ksw_pk<(anonymous namespace)::PipeFetch, 5>(...)": the template arguments contain parentheses themselves
    m = re.search(r"(k_\w+)(<.*?, (\d)>)?\(", n.replace("(anonymous namespace)::", ""))
    name = (m.group(1) + ("<%s>" % m.group(3) if m and m.group(3) else "")) if m else "scan/prim"
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
if not rows:
    sys.exit("no kernels")
# skip the index build: start at the first k_seed
t_first = min(s for s, e, n, q, st in rows if n.startswith("k_seed"))
rows = [x for x in rows if x[0] >= t_first]
ev = []
for s, e, n, q, st in rows:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
busy = {}; cur = 0; last = ev[0][0]
active = {}
pair = {}
for t, d, n in ev:
    busy[cur] = busy.get(cur, 0) + (t - last)
    if cur >= 2:
        key = "+".join(sorted(k for k, v in active.items() if v > 0))
        pair[key] = pair.get(key, 0) + (t - last)
    last = t; cur += d; active[n] = active.get(n, 0) + d
wall = ev[-1][0] - ev[0][0]
print("wall %.1f ms" % (wall / 1e6))
for k in sorted(busy):
    print("  %d kernels in flight: %.1f ms (%.0f%%)" % (k, busy[k] / 1e6, 100.0 * busy[k] / wall))
tot = {}
cnt = {}
for s, e, n, q, st in rows:
    tot[n] = tot.get(n, 0) + (e - s); cnt[n] = cnt.get(n, 0) + 1
for n in sorted(tot, key=lambda k: -tot[k])[:14]:
    print("  %-22s %4d launches, mean %.3f ms, total %.1f ms" % (n, cnt[n], tot[n] / cnt[n] / 1e6, tot[n] / 1e6))
print("most common concurrent sets:")
for k in sorted(pair, key=lambda k: -pair[k])[:12]:
    print("  %6.1f ms  %s" % (pair[k] / 1e6, k))
print("queues:", sorted(set(q for s, e, n, q, st in rows)), "streams:", sorted(set(st for s, e, n, q, st in rows)))
