#!/usr/bin/env python3
"""Phase cycle counters of k_chain (needs `make -C ma_amd/csrc chainprof`; run with MA_LANES_PER_WAVE=1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MA_AMD_LIB"] = os.path.join(ROOT, "tools", "_prof", "libma_amd_chainprof.so")
os.environ.setdefault("MA_LANES_PER_WAVE", "1")
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--boundary-reads", "0", "--overlap", "0"] + sys.argv[1:]
import runpy
import ma_amd
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    out = (C.c_ulonglong * 16)()
    ma_amd.lib().ma_debug_chain_prof(out)
    v = list(out)
    names = ["sort by delta", "window sweep", "heap + rectangles", "sort by ref", "strip ranges", "harm: 2 medians", "harm: ransac",
             "harm: outliers + linesweeps", "harm: final sort", "chain_read total"]
    reads = max(v[10], 1)
    print("reads %d, SoC tries/read %.2f, seeds/read %.1f" % (v[10], v[11] / reads, v[12] / reads), file=sys.stderr)
    for n, x in zip(names, v):
        print("%-28s %12.0f cycles/read  %5.1f %%" % (n, x / reads, 100.0 * x / max(v[9], 1)), file=sys.stderr)
