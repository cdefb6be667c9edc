#!/usr/bin/env python3
"""What became of the extension jobs tried on the proven narrow band (ksw_band.h, MA_KSW_GRP=3) during a bench.py run:
   usage (GPU box): MA_KSW_GRP=3 python tools/band_stats.py --workload 150bp --overlap 0 --boundary-reads 0"""
import ctypes as C, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"] + sys.argv[1:]
import ma_amd
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    out = (C.c_ulonglong * 8)()
    ma_amd.lib().ma_debug_band_stats(out)
    v = list(out)
    n = max(v[0], 1)
    print("narrow band: %d jobs tried, %d proved (%.1f %%); failed check 1 (maximum vs cells outside the band) %d, 2 (class maxima) %d, "
          "3 (back-trace start) %d, 4 (z-drop) %d; handed back for another reason %d; %.1f diagonals per job" % (
              v[0], v[1], 100.0 * v[1] / n, v[2], v[3], v[4], v[5], v[6], v[7] / n), file=sys.stderr)
