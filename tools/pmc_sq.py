#!/usr/bin/env python3
"""Sums SQ counters per kernel from a rocprofv3 --pmc csv (diagnostics).  usage: pmc_sq.py <dir> [substring ...]"""
import csv, glob, re, sys, collections
d = sys.argv[1]
pats = sys.argv[2:] or ["k_ksw"]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not any(p in n for p in pats):
        continue
    m = re.search(r"(k_\w+)(<[^>]*?, (\d)>)?", n.replace("(anonymous namespace)::", ""))
    short = m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[short].add(r["Dispatch_Id"])
for k, v in acc.items():
    print(k, "dispatches", len(calls[k]))
    for c, x in sorted(v.items()):
        print("   %-24s %.4g" % (c, x))
