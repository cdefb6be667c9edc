#!/usr/bin/env python3
"""Concurrency timeline of a rocprofv3 --kernel-trace run of ONE overlapped leg of bench.py (--inflight N): which kernels
run side by side, what a kernel's launch costs under overlap against the same kernel alone (the warm-up steps run one
batch at a time), how much of the timed region the GPU idles.
usage: overlap_timeline.py <kernel_trace.csv> <batches in flight> <warm-up steps per batch> [--steps-ms <ms_per_step of the leg>]"""
import csv
import re
import sys


def kname(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"(?:\w+::)*(\w+)", n)
    base = m.group(1) if m else n[:24]
    if base.startswith("k_ksw"):
        t = re.search(r",\s*(\d+)\s*>\s*\(", n)
        if t:
            base += "<%s>" % t.group(1)
    if not base.startswith("k_"):
        base = "rocprim/fill"
    return base


def main():
    path, nb, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"]), r.get("Stream_Id", "?")))
    rows.sort()
    seeds = [x for x in rows if x[2] in ("k_seed", "k_seed_long", "k_seed_tasks")]
    if len(seeds) <= nb * warm:
        sys.exit("not enough k_seed launches (%d) for %d x %d warm-up steps" % (len(seeds), nb, warm))
    t0 = seeds[nb * warm][0]  # first timed step
    t_warm0 = seeds[0][0]
    t1 = max(e for s, e, n, st in rows)
    timed = [x for x in rows if x[0] >= t0]
    alone = [x for x in rows if t_warm0 <= x[0] < t0]
    n_steps = len([x for x in timed if x[2] in ("k_seed", "k_seed_long", "k_seed_tasks")])
    wall = t1 - t0
    print("timed region: %.1f ms, %d steps on %d streams (%.2f ms per step); warm-up region (one batch at a time): %d steps" % (
        wall / 1e6, n_steps, len(set(x[3] for x in timed)), wall / 1e6 / max(n_steps, 1), len(seeds) - n_steps))
    ev = []
    for s, e, n, st in timed:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort()
    busy, sets, active, cur, last = {}, {}, {}, 0, t0
    for t, d, n in ev:
        busy[cur] = busy.get(cur, 0) + (t - last)
        key = "+".join(sorted(k for k, v in active.items() if v > 0)) or "(idle)"
        sets[key] = sets.get(key, 0) + (t - last)
        last, cur = t, cur + d
        active[n] = active.get(n, 0) + d
    print("kernels running at once (share of the timed region):")
    for k in sorted(busy):
        print("  %d: %7.1f ms  %5.1f %%" % (k, busy[k] / 1e6, 100.0 * busy[k] / wall))

    def stats(rs):
        tot, cnt = {}, {}
        for s, e, n, st in rs:
            tot[n] = tot.get(n, 0) + (e - s)
            cnt[n] = cnt.get(n, 0) + 1
        return tot, cnt
    tt, tc = stats(timed)
    at, ac = stats(alone)
    a_steps = max(len(seeds) - n_steps, 1)
    print("per step and kernel: ms under overlap | ms alone (warm-up steps) | ratio")
    so = sa = 0.0
    for n in sorted(tt, key=lambda k: -tt[k]):
        o = tt[n] / 1e6 / max(n_steps, 1)
        a = at.get(n, 0) / 1e6 / a_steps
        so, sa = so + o, sa + a
        if o >= 0.05:
            print("  %-16s %8.2f | %8.2f | %5.2f   (%d launches per step)" % (n, o, a, o / a if a > 0 else float("nan"), round(tc[n] / max(n_steps, 1))))
    print("  %-16s %8.2f | %8.2f | %5.2f   sum over kernels; wall per step %.2f -> overlap hides %.0f %% of the summed kernel time" % (
        "total", so, sa, so / sa if sa else float("nan"), wall / 1e6 / max(n_steps, 1), 100.0 * (1 - wall / 1e6 / max(n_steps, 1) / so) if so else 0))
    print("most common concurrent sets:")
    for k in sorted(sets, key=lambda k: -sets[k])[:14]:
        print("  %7.1f ms %5.1f %%  %s" % (sets[k] / 1e6, 100.0 * sets[k] / wall, k))


if __name__ == "__main__":
    main()
