#!/usr/bin/env python3
"""Randomised soak of the exact kswcpp kernels (ma_ksw_batch: k_ksw_pk<1..5>, k_ksw) against the oracle, every ez field and
the cigar: many seeds, short and long cases, bands that cut the rectangle, N bases, all scoring defaults.
usage: python tools/ksw_soak.py [seeds=12] > gpurun_out/ksw_soak.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ma_amd  # noqa: E402
from ma_testlib import or_ksw, or_params, rand_ksw_cases  # noqa: E402
from test_gpu_round2 import band_cut_extension_cases  # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    P = ma_amd.Params.preset("default")
    op = or_params()
    total = bad = 0
    for s in range(seeds):
        cases = rand_ksw_cases(1200, 9000 + s, max_len=300) + rand_ksw_cases(12, 9500 + s, long_frac=1.0) + band_cut_extension_cases(60, 9700 + s)
        rng = np.random.default_rng(9900 + s)
        for _ in range(40):  # sequences with several N bases, both directions
            ql, tl = int(rng.integers(20, 900)), int(rng.integers(20, 900))
            q = rng.integers(0, 5, size=ql, dtype=np.uint8)
            t = rng.integers(0, 5, size=tl, dtype=np.uint8)
            cases.append((q, t, int(rng.choice([40, 200, 512])), int(rng.choice([-1, 100, 200])), int(rng.choice([0, 0x40, 0x40 | 0x02 | 0x80]))))
        ez, cigs = ma_amd.ksw_batch(P, cases)
        for i, (q, t, w, zd, fl) in enumerate(cases):
            oez, ocig = or_ksw(op, q, t, w, zd, fl)
            ok = all(int(ez[f][i]) == int(oez[f]) for f in oez.dtype.names) and np.array_equal(cigs[i], ocig)
            total += 1
            if not ok:
                bad += 1
                print("MISMATCH seed %d case %d qlen %d tlen %d w %d zdrop %d flag %d" % (s, i, len(q), len(t), w, zd, fl), flush=True)
        print("seed %d: %d cases so far, %d mismatching" % (s, total, bad), flush=True)
    print("total %d cases, %d mismatching" % (total, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
