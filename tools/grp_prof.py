#!/usr/bin/env python3
"""Phase cycle counters of k_ksw_grp, the extension kernel that runs several short jobs per wavefront (ksw_grp.h; needs the
-DMA_KSW_PROF build: make -C ma_amd/csrc prof).  usage: python tools/grp_prof.py --workload 150bp --overlap 0 --boundary-reads 0"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MA_AMD_LIB"] = os.path.join(ROOT, "tools", "_prof", "libma_amd_prof.so")
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "2", "--warmup", "0", "--cpu-sample", "0"] + sys.argv[1:]
import runpy
import ma_amd
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    out = (C.c_ulonglong * 24)()
    ma_amd.lib().ma_debug_grp_prof(out)
    v = list(out)
    for k, G in enumerate((1, 2, 4)):
        p = v[8 * k:8 * k + 8]
        if not p[5]:
            continue
        tot = sum(p[:4])
        print("G = %d: %d sets, %d jobs (%.2f per set); wave cycles per set: set-up %.0f, diagonal loop %.0f, position + back-trace %.0f, "
              "publish %.0f (total %.0f); diagonals per set %.1f, the jobs' own %.1f on average (lock-step waste %.2fx); loop cycles per "
              "diagonal %.0f; share of all k_ksw_grp wave cycles %.1f %%" % (
                  G, p[5], p[6], p[6] / p[5], p[0] / p[5], p[1] / p[5], p[2] / p[5], p[3] / p[5], tot / p[5], p[4] / p[5], p[7] / max(p[6], 1),
                  p[4] * p[6] / max(p[5], 1) / max(p[7], 1), p[1] / max(p[4], 1), 100.0 * tot / max(sum(v[8 * i + j] for i in range(3) for j in range(4)), 1)))
