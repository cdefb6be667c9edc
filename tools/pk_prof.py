#!/usr/bin/env python3
"""Wave cycles per phase of one diagonal of k_ksw_pk<5> (needs the -DMA_KSW_PROF build: make -C ma_amd/csrc prof).
usage: python tools/pk_prof.py --workload 10kb [bench.py options]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MA_AMD_LIB"] = os.path.join(ROOT, "tools", "_prof", "libma_amd_prof.so")
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--overlap", "0", "--boundary-reads", "0"] + sys.argv[1:]
import runpy
import ma_amd
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    out = (C.c_ulonglong * 16)()
    ma_amd.lib().ma_debug_pk_prof(out)
    v = list(out)
    nd, jobs = max(v[6], 1), max(v[10], 1)
    names = ["bounds+rotation+prologue", "neighbour views + query bases", "slot bodies", "H[en0] pick", "raise/snapshot/z-drop need",
             "mte/mqe/z-drop/early bound/tail"]
    tot = sum(v[:6])
    print("k_ksw_pk<5>: %d jobs, %d diagonals (%.0f per job), %.2f active slots per diagonal, raised on %.1f %%, exact max on %.2f %% of the diagonals" % (
        v[10], v[6], v[6] / jobs, v[7] / nd, 100.0 * v[8] / nd, 100.0 * v[9] / nd))
    for n, x in zip(names, v[:6]):
        print("  %-34s %8.0f cycles per diagonal  %5.1f %%" % (n, x / nd, 100.0 * x / max(tot, 1)))
    print("  %-34s %8.0f cycles per diagonal" % ("sum", tot / nd))
    print("  back-trace + position of the maximum: %.0f cycles per job (the loop: %.0f)" % (v[11] / jobs, tot / jobs))
