#!/usr/bin/env python3
"""Phase cycle counters of the extension kernel (needs the -DMA_KSW_PROF build, see tools/_prof)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MA_AMD_LIB"] = os.path.join(ROOT, "tools", "_prof", "libma_amd_prof.so")
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "2", "--warmup", "0", "--cpu-sample", "0"] + sys.argv[1:]
import runpy
import ma_amd
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    out = (C.c_ulonglong * 16)()
    ma_amd.lib().ma_debug_ksw_prof(out)
    v = list(out)
    names = ["fetch+setup", "core(total)", "publish", "jobs", "ext:loop", "ext:backtrace", "ext:cells", "ext:jobs",
             "glob:loop", "glob:backtrace", "glob:cells", "glob:jobs"]
    for n, x in zip(names, v):
        print("%-16s %d" % (n, x))
    sp = (C.c_ulonglong * 8)()
    ma_amd.lib().ma_debug_seed_prof(sp)
    sp = list(sp)
    slow_trips = sp[5] >> 32
    print("k_seed: slow-path cycles %d in %d slow trips (%.0f each), try+prepare %d, extend+apply %d, wave trips %d, active lanes/trip %.1f, refill trips %d" % (
        sp[0], slow_trips, sp[0] / max(slow_trips, 1), sp[1], sp[2], sp[3], sp[4] / max(sp[3], 1), sp[5] & 0xffffffff))
    print("k_seed per trip: refill %.0f prepare %.0f extend %.0f" % (sp[0] / max(sp[3], 1), sp[1] / max(sp[3], 1), sp[2] / max(sp[3], 1)))
    print("k_extract: wave cycles in segment search %d, in bwt_sa %d" % (sp[6], sp[7]))
    j = max(v[3], 1)
    print("per job cycles: fetch %.0f core %.0f publish %.0f" % (v[0] / j, v[1] / j, v[2] / j))
    if v[7]:
        print("ext  per job: loop %.0f backtrace %.0f cells %.0f" % (v[4] / v[7], v[5] / v[7], v[6] / v[7]))
    if v[7]:
        print("ext  per job: descriptor+query %.0f target+init %.0f" % (v[12] / v[7], v[13] / v[7]))
    print("global (cigar-only) jobs: %d" % v[8])
    if v[9]:
        print("extension kernel wave time (fetch .. publish): all jobs %d cycles; jobs that fit half a wavefront (qlen + 2 <= 64): "
              "%d jobs = %.1f %% of the jobs, %.1f %% of the time; jobs that fit a quarter (qlen + 2 <= 32): %d jobs = %.1f %%, %.1f %% of the time" % (
                  v[9], v[15], 100.0 * v[15] / j, 100.0 * v[14] / v[9], v[11], 100.0 * v[11] / j, 100.0 * v[10] / v[9]))
        half_only = v[14] - v[10]
        print("upper bound of packing: two jobs per wave for the half-size ones saves at most %.1f %% of the kernel; four per wave for "
              "the quarter-size ones on top of that at most %.1f %% more (perfect pairing, no added per-diagonal cost)" % (
                  50.0 * v[14] / v[9], 25.0 * v[10] / v[9]))
