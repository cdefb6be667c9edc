#!/usr/bin/env python3
"""CPU experiment (round 6): which share of the LONG extension jobs (query > 254 bases) of the 10 kb and 50 kb
workloads could a narrow band of B cells either side of the main diagonal PROVE?  The jobs have the shape the pipeline emits
(needlemanWunsch.cpp:708-716, 781-782: the rest of the read against the seed's neighbourhood padded by 1000 reference bases; band 512,
z-drop 200).  For every job the oracle's kswcpp runs at the full band; a band B can prove it when
  (1) ez.max > match * min(qlen, tlen) - f(B + 1)      (no cell outside the band reaches the maximum; f = cheaper gap cost), and
  (2) the optimal path (the cigar from (max_t, max_q)) stays within |t - j| <= B.
usage: python tools/band_long_experiment.py [jobs=150]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ma_testlib import KSW_EXTZ, KSW_REV, KSW_RIGHT, or_ksw, or_params  # noqa: E402


def noisy(ref, n, sub, ins, dele, rng):
    out, i = [], 0
    while len(out) < n and i < len(ref):
        u = rng.random()
        if u < sub:
            out.append((int(ref[i]) + 1 + int(rng.integers(0, 3))) % 4); i += 1
        elif u < sub + ins:
            out.append(int(rng.integers(0, 4)))
        elif u < sub + ins + dele:
            i += 1
        else:
            out.append(int(ref[i])); i += 1
    return np.array(out[:n], dtype=np.uint8)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    op = or_params("default", 1)
    a, q1, e1, q2, e2 = 2, 4, 2, 24, 1
    f = lambda L: min(q1 + L * e1, q2 + L * e2)  # noqa: E731
    bands = [24, 48, 64, 96, 120]
    for name, ql, tl, er in (("10 kb end extension (2500 x 1000, 0.4/0.3/0.3 %)", 2500, 1000, (0.004, 0.003, 0.003)),
                             ("10 kb dual extension (600 x 640)", 600, 640, (0.004, 0.003, 0.003)),
                             ("50 kb gap extension (1058 x 375, 3/3/4 %)", 1058, 375, (0.03, 0.03, 0.04))):
        rng = np.random.default_rng(5)
        ok = {b: 0 for b in bands}
        fail1 = {b: 0 for b in bands}
        lost_all = []
        for k in range(n):
            ref = rng.integers(0, 4, size=ql + tl + 400, dtype=np.uint8)
            q = noisy(ref, ql, *er, rng)
            if rng.random() < 0.7:
                q[0] = (q[0] + 1) % 4  # a seed ended here
            t = np.ascontiguousarray(ref[:tl])
            fl = KSW_EXTZ if k % 2 == 0 else (KSW_EXTZ | KSW_RIGHT | KSW_REV)
            ez, cig = or_ksw(op, q, t, 512, 200, fl)
            mx = int(ez["max"])
            # the path's largest offset from the main diagonal
            ops = [(int(c) & 0xf, int(c) >> 4) for c in cig]
            if fl & KSW_REV:
                ops = ops[::-1]
            off, worst = 0, 0
            for o, l in ops:
                if o == 1:
                    off -= l
                elif o == 2:
                    off += l
                worst = max(worst, abs(off))
            lost = a * min(len(q), tl) - mx
            lost_all.append(lost)
            for b in bands:
                c1 = mx > a * min(len(q), tl) - f(b + 1)
                fail1[b] += 0 if c1 else 1
                ok[b] += 1 if (c1 and worst <= b) else 0
        la = np.array(lost_all)
        print("%s: %d jobs, points lost against match * min(qlen, tlen): mean %.1f, median %.0f, 90 %% %.0f, max %d" % (
            name, n, la.mean(), np.median(la), np.percentile(la, 90), la.max()))
        for b in bands:
            print("   B = %3d (G = %3d): provable %5.1f %%   (fail check 1: %5.1f %%)" % (b, f(b + 1), 100.0 * ok[b] / n, 100.0 * fail1[b] / n))


if __name__ == "__main__":
    main()
