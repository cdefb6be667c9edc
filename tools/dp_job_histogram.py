#!/usr/bin/env python3
"""Shapes of the kswcpp calls of a batch (ma_batch_get_dp_jobs): how many jobs, anti-diagonals and cells fall into which query-length
class, global (gap fills between seeds) and extension jobs apart.   usage (GPU box): python tools/dp_job_histogram.py [read_len n_reads sub ins dele]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ma_amd
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616,
          64444167, 46709983, 50818468, 156040895, 57227415]
rl = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sub, ins, dele = (float(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])) if len(sys.argv) > 5 else (0.03, 0.03, 0.04)
ma_amd.set_device(0)
L = ma_amd.lib()
lens = np.array(GRCH38, dtype=np.uint64)
F = int(lens.sum())
g = torch.empty(F, dtype=torch.uint8, device="cuda")
assert L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(F), C.c_int32(1), C.c_void_p(g.data_ptr())) == 0
idx = ma_amd.Index.build_device(lens, g.data_ptr())
del g
cap = int(n * (rl * (1 + 2 * ins) + 8)) + 1024
codes = torch.empty(cap, dtype=torch.uint8, device="cuda")
offs = torch.empty(n + 1, dtype=torch.int64, device="cuda")
nb = C.c_uint64()
assert L.ma_synth_reads_device(idx.h, C.c_uint64(13), C.c_uint64(n), C.c_uint32(rl), C.c_double(sub), C.c_double(ins), C.c_double(dele),
                               C.c_uint64(0), C.c_void_p(codes.data_ptr()), C.c_void_p(offs.data_ptr()), C.c_uint64(cap), C.byref(nb)) == 0
bt = ma_amd.Batch(idx, ma_amd.Params.preset("default"), n, int(nb.value) + 64)
bt.set_reads_device(codes.data_ptr(), offs.data_ptr(), n, int(nb.value))
bt.align()
bt.sync()
J = bt.dp_jobs()
q, t, w, zd, fl = (J[:, i].astype(np.int64) for i in range(5))
ext = (fl & 0x40) != 0
print("%d reads of %d bp (%.1f/%.1f/%.1f %%): %d kswcpp calls" % (n, rl, 100 * sub, 100 * ins, 100 * dele, len(q)))
edges = [0, 1, 8, 16, 32, 64, 126, 254, 1 << 30]
right = (fl & 0x02) != 0
for kind, m in (("global", ~ext), ("extension", ext), ("extension, left-aligned (read ends)", ext & ~right), ("extension, right-aligned and reversed (read starts)", ext & right)):
    print("%s jobs: %d" % (kind, int(m.sum())))
    for lo, hi in zip(edges[:-1], edges[1:]):
        s = m & (q > lo) & (q <= hi)
        if not s.any():
            continue
        diag = (q[s] + t[s] - 1)
        cut = np.minimum(diag, np.where(ext[s], 2 * q[s] + 16, diag))  # an extension stops soon after the query's end
        cells = np.where(ext[s], q[s] * np.minimum(t[s], q[s] + 16), q[s] * t[s])
        print("  qlen %4d..%-10d %9d jobs  mean q %6.1f t %7.1f w %6.1f  diagonals %12d  cells %14d" % (
            lo + 1, hi, int(s.sum()), q[s].mean(), t[s].mean(), w[s].mean(), int(cut.sum()), int(cells.sum())))

# the long extension jobs (queries beyond 254 bases) by shape: which of them is an end extension (target = the padding) and which a
# two-sided extension into a gap (target about as long as the query) -- what the band of 120 (ksw_band.h, G = 1) is planned for
lg = ext & (q > 254)
if lg.any():
    mn = np.minimum(q[lg], t[lg])
    print("long extension jobs: %d; min(qlen, tlen) quantiles 50 / 90 / 99 / 100 %%: %s" % (int(lg.sum()), np.percentile(mn, [50, 90, 99, 100]).astype(int).tolist()))
    for lo, hi in ((0, 1100), (1100, 2048), (2048, 4096), (4096, 7900), (7900, 1 << 30)):
        s2 = (mn > lo) & (mn <= hi)
        if s2.any():
            ql, tl = q[lg][s2], t[lg][s2]
            print("  min(qlen, tlen) %5d..%-10d %8d jobs  mean q %7.1f t %7.1f  |q - t| < 120: %5.1f %%  diagonals (2 min + 120) %12d" % (
                lo + 1, hi, int(s2.sum()), ql.mean(), tl.mean(), 100.0 * float((np.abs(ql - tl) < 120).mean()), int((2 * mn[s2] + 120).sum())))
