#!/usr/bin/env python3
"""Distribution of the direction-matrix scratch (ksw_p_bytes) of the exact-kernel DP jobs of one bench step, by kernel class:
jobs and upper-bound work (diagonals x ring slots) per power-of-two bucket.  Diagnostics for the launch planning."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GRCH38


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--sub", type=float, default=0.004)
    ap.add_argument("--ins", type=float, default=0.003)
    ap.add_argument("--dele", type=float, default=0.003)
    a = ap.parse_args()
    import torch, ma_amd
    dev = torch.device("cuda", 0)
    L = ma_amd.lib()
    lens = np.array(GRCH38, dtype=np.uint64)
    F = int(lens.sum())
    g = torch.empty(F, dtype=torch.uint8, device=dev)
    assert L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(F), C.c_int32(1), C.c_void_p(g.data_ptr())) == 0
    idx = ma_amd.Index.build_device(lens, g.data_ptr())
    del g
    n = a.reads
    cap = int(n * (a.read_len * (1 + 2 * a.ins) + 8)) + 1024
    codes = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nb = C.c_uint64()
    assert L.ma_synth_reads_device(idx.h, C.c_uint64(11), C.c_uint64(n), C.c_uint32(a.read_len), C.c_double(a.sub),
                                   C.c_double(a.ins), C.c_double(a.dele), C.c_uint64(0), C.c_void_p(codes.data_ptr()),
                                   C.c_void_p(offs.data_ptr()), C.c_uint64(cap), C.byref(nb)) == 0
    P = ma_amd.Params.preset("default")
    b = ma_amd.Batch(idx, P, n, int(nb.value) + 64)
    b.set_reads_device(codes.data_ptr(), offs.data_ptr(), n, int(nb.value))
    b.align()
    b.sync()
    J = b.dp_jobs().astype(np.int64)
    ql, tl, w, zd, fl, zdr, mq, mt = J.T
    weff = np.where(w < 0, np.maximum(ql, tl), w)
    m = np.minimum(np.minimum(ql, tl), weff + 1)
    slots = (m + 30 + 127) // 128
    # n_col of kswcpp: band cells rounded to 16-lane blocks (upper bound used by ksw_p_bytes)
    ncol = (np.minimum(np.minimum(tl, ql), weff + 1) + 15) // 16 + 1
    p = (ql + tl - 1) * ncol * 16 + 16
    # diagonals actually run: extension jobs stop at max + early stop; upper bound qlen+tlen-1, proxy 2*min+|band|
    ext = (fl & 0x40) != 0
    small = (ql <= 254) & ext  # roughly the extension kernel's jobs
    print("jobs", len(J), "ext-kernel-like", int(small.sum()))
    for cls, lo, hi in (("pk<1>", 1, 1), ("pk<2>", 2, 2), ("pk<3>", 3, 3), ("pk<5>", 4, 5), ("lds", 6, 1 << 30)):
        k = (slots >= lo) & (slots <= hi) & ~small
        if not k.any():
            continue
        work = ((ql + tl - 1) * slots)[k]
        pk = p[k]
        print("%s: jobs %d, work %.3g slot-diagonals, p max %.1f MB, qlen max %d" % (cls, k.sum(), work.sum(), pk.max() / 1e6, ql[k].max()))
        edges = [0] + [1 << s for s in range(16, 30)]
        for a_, b_ in zip(edges[:-1], edges[1:]):
            kk = (pk > a_) & (pk <= b_)
            if kk.any():
                print("   p in (%8.2f, %8.2f] MB: jobs %7d (%5.1f%%)  work %5.1f%%  qlen mean %7.0f max %6d  tlen mean %7.0f" % (
                    a_ / 1e6, b_ / 1e6, kk.sum(), 100 * kk.mean(), 100 * work[kk].sum() / work.sum(), ql[k][kk].mean(), ql[k][kk].max(), tl[k][kk].mean()))


if __name__ == "__main__":
    main()
