#!/usr/bin/env python3
"""Overlap matrix on one workload, ONE process (the index is built once): batches in flight x DP turn-taking
(MA_DP_EXCLUSIVE: one DP stage at a time per device) x resident DP waves per CU (MA_KSW_WAVES_PER_CU: fewer persistent DP
waves leave register file and wave slots to the memory-bound kernels of the other batches).
(The CU-masked streams of round 5 -- every split slower than none, profiles/r05_overlap_matrix_150bp_cu_split.txt -- were taken
out of the library in round 6.)
usage: python tools/overlap_matrix.py [--workload 150bp] [--steps 12] [--host-io 0] > gpurun_out/overlap_matrix.txt"""
import argparse
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="150bp")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--host-io", type=int, default=0)
    ap.add_argument("--inflight", default="1,2,3")
    ap.add_argument("--waves", default="0,12,16,20,24")
    ap.add_argument("--exclusive", default="0,1")
    ap.add_argument("--seed-lanes", default="0", help="MA_SEED_LANES values (0 = default: a lane per read up to the resident lanes)")
    a = ap.parse_args()
    args = bench.build_parser().parse_args(["--workload", a.workload, "--cpu-sample", "0", "--boundary-reads", "0"])
    E = bench.Env(args)
    wl = dict(bench.WORKLOADS[a.workload])
    wl["warmup"] = 1
    print("workload %s, %d steps per point, host_io %d; columns: reads/s, ms per step, k_ksw ms per step under that overlap" % (
        a.workload, a.steps, a.host_io), flush=True)
    import itertools
    ints = lambda t: [int(x) for x in t.split(",")]  # noqa: E731
    for nfl, ex, sl, wv in itertools.product(ints(a.inflight), ints(a.exclusive), ints(a.seed_lanes), ints(a.waves)):
        if nfl == 1 and ex == 1:
            continue
        for key, val in (("MA_SEED_LANES", sl), ("MA_KSW_WAVES_PER_CU", wv)):
            if val:
                os.environ[key] = str(val)
            else:
                os.environ.pop(key, None)
        os.environ["MA_DP_EXCLUSIVE"] = str(ex)
        a2 = copy.copy(args)
        a2.inflight, a2.host_io, a2.cpu_sample = nfl, a.host_io, 0
        w2 = dict(wl)
        w2["steps"] = max(a.steps, 3 * nfl)
        r = bench.run_workload(E, a.workload, w2, a2)
        k = r["roofline"]["kernel_ms_per_step"]
        print("inflight=%d dp_exclusive=%d seed_lanes=%s dp_waves_per_cu=%s  %12.1f  %8.3f  k_ksw %.2f k_seed %.2f k_chain %.2f" % (
            nfl, ex, sl or "default", wv or "default", r["value"], r["ms_per_step"],
            k["k_ksw"], k["k_seed"], k["k_chain"]), flush=True)
    E.close()


if __name__ == "__main__":
    main()
