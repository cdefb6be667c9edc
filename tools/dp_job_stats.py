#!/usr/bin/env python3
"""Histogram of the kswcpp calls one bench step issues (diagnostics for kernel tuning; not part of the product)."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GRCH38


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--preset", default="default")
    ap.add_argument("--sub", type=float, default=0.005)
    ap.add_argument("--ins", type=float, default=0.0)
    ap.add_argument("--dele", type=float, default=0.0)
    a = ap.parse_args()
    import torch, ma_amd
    dev = torch.device("cuda", 0)
    L = ma_amd.lib()
    lens = np.array([max(1000, int(x * a.genome_scale)) for x in GRCH38], dtype=np.uint64)
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
    P = ma_amd.Params.preset(a.preset)
    b = ma_amd.Batch(idx, P, n, int(nb.value) + 64)
    b.set_reads_device(codes.data_ptr(), offs.data_ptr(), n, int(nb.value))
    b.align()
    b.sync()
    J = b.dp_jobs().astype(np.int64)
    c = b.counts()
    print("reads", n, "counts", c)
    print("jobs", len(J), "per read", len(J) / n)
    ql, tl, w, zd, fl, zdr, mq, mt = J.T
    weff = np.where(w < 0, np.maximum(ql, tl), w)
    m = np.minimum(np.minimum(ql, tl), weff + 1)
    diags = ql + tl - 1
    print("total diags (upper bound)", diags.sum(), "per read", diags.sum() / n)
    for name, v in (("qlen", ql), ("tlen", tl), ("w", w), ("m", m), ("diags", diags)):
        print(name, "pct 10/50/90/99/max", np.percentile(v, [10, 50, 90, 99, 100]).tolist(), "mean", v.mean())
    print("flag hist", dict(zip(*np.unique(fl, return_counts=True))))
    print("zdrop hist", dict(zip(*np.unique(zd, return_counts=True))))
    print("zdropped frac", zdr.mean())
    for lo, hi in ((0, 2), (2, 34), (34, 98), (98, 162), (162, 546), (546, 1 << 30)):
        k = (m > lo) & (m <= hi)
        print("m in (%d,%d]: jobs %d (%.1f%%), diags %d (%.1f%%), mean ql %.1f tl %.1f" % (
            lo, hi, k.sum(), 100 * k.mean(), diags[k].sum(), 100 * diags[k].sum() / max(1, diags.sum()),
            ql[k].mean() if k.any() else 0, tl[k].mean() if k.any() else 0))
    # 2-D histogram of (qlen, tlen) buckets weighted by diagonals
    qb = np.minimum(ql // 16, 12)
    tb = np.minimum(tl // 16, 24)
    H = np.zeros((13, 25))
    np.add.at(H, (qb, tb), diags)
    H = 100 * H / H.sum()
    print("diag share by qlen//16 (rows) x tlen//16 (cols)")
    for r in range(13):
        print(" ".join("%5.1f" % x for x in H[r]))
    for nm, k in (("global", (fl & 0x40) == 0), ("ext", (fl & 0x40) != 0)):
        if k.any():
            print(nm, "jobs", int(k.sum()), "qlen pct", np.percentile(ql[k], [10, 50, 90, 99, 100]).tolist(), "tlen pct",
                  np.percentile(tl[k], [10, 50, 90, 99, 100]).tolist(), "w pct", np.percentile(w[k], [10, 50, 90, 100]).tolist(),
                  "diag share %.3f" % (diags[k].sum() / diags.sum()))
            elig = k & (ql <= weff + 1) & (ql <= 256) & ((ql + 15 <= 256) | (tl <= 256)) & (((fl & 0x40) != 0) | (ql + tl - 2 <= weff))
            print("   eligible for the extension kernel: %.3f of jobs, %.3f of their diagonals" % (elig.sum() / k.sum(), diags[elig].sum() / max(1, diags[k].sum())))
    ext = (fl & 0x40) != 0
    print("extension jobs (EXTZ_ONLY)", ext.mean(), "diag share", diags[ext].sum() / diags.sum())
    e = zdr == 1
    print("zdropped jobs: mean max_q+max_t", (mq[e] + mt[e]).mean() if e.any() else 0, "mean diags", diags[e].mean() if e.any() else 0)


if __name__ == "__main__":
    main()
