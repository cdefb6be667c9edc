#!/usr/bin/env python3
"""Randomised soak of the proven narrow band (ksw_band.h) against the oracle's kswcpp at the full band: max, max_q, max_t and the
cigar of every job, proved or handed on -- the short jobs (four per wavefront, B = 24) and, since round 6, the long ones (one per
wavefront, B = 120) and the adversarial generator of tests/test_gpu_round6.py (an out-of-band path within a few points of the in-band
optimum, one gap along the band's edge, tiny z-drops) under three scoring schemes.
usage (GPU box): python tools/band_soak.py [seeds=10] > gpurun_out/band_soak.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ma_amd  # noqa: E402
from ma_testlib import KSW_EXTZ, KSW_REV, KSW_RIGHT, or_ksw, or_params  # noqa: E402
from test_gpu_round5 import band_extension_cases  # noqa: E402
from test_gpu_round6 import SCORINGS, adversarial_cases, long_extension_cases, params_for  # noqa: E402


def read_end_cases(n, seed):
    """what a 150 bp batch produces: the read continues on the reference with 0.5 .. 3 % substitutions behind a first mismatch,
    sometimes with one short indel; target = query + 1000 padded bases"""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        ql = int(rng.integers(33, 151))
        t = rng.integers(0, 4, size=ql + 1000, dtype=np.uint8)
        if rng.random() < 0.2:
            unit = rng.integers(0, 4, size=int(rng.integers(1, 9)), dtype=np.uint8)
            s0, L = int(rng.integers(0, ql)), int(rng.integers(8, 80))
            t[s0:s0 + L] = np.resize(unit, L)
        q = t[:ql].copy()
        mut = rng.random(ql) < rng.choice([0.005, 0.01, 0.03])
        q[mut] = (q[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
        q[0] = (q[0] + 1) % 4
        if rng.random() < 0.15:
            p, g = int(rng.integers(5, ql - 5)), int(rng.integers(1, 6))
            q = np.concatenate([q[:p], q[p + g:], t[ql:ql + g]]) if rng.random() < 0.5 else np.concatenate([q[:p], rng.integers(0, 4, size=g, dtype=np.uint8), q[p:]])[:ql]
        fl = KSW_EXTZ if rng.random() < 0.5 else (KSW_EXTZ | KSW_RIGHT | KSW_REV)
        cases.append((np.ascontiguousarray(q, dtype=np.uint8), t, 512, 200, fl))
    return cases


def stats(long_jobs=False):
    out = (C.c_ulonglong * 8)()
    (ma_amd.lib().ma_debug_band_long_stats if long_jobs else ma_amd.lib().ma_debug_band_stats)(out)
    return np.array(list(out), dtype=np.int64)


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ma_amd.set_device(0)
    P = ma_amd.Params.preset("default")
    op = or_params()
    total = bad = 0
    s0 = stats()
    for s in range(seeds):
        for every in (True, False):
            os.environ["MA_KSW_GRP"] = "1033"
            if every:
                os.environ["MA_KSW_BAND_ALL"] = "1"
            else:
                os.environ.pop("MA_KSW_BAND_ALL", None)
            cases = band_extension_cases(2500, 7000 + s, qmin=33) + read_end_cases(2500, 7500 + s)
            ez, cigs = ma_amd.ksw_batch(P, cases, pipeline_semantics=True)
            for i, (q, t, w, zd, fl) in enumerate(cases):
                oez, ocig = or_ksw(op, q, t, w, zd, fl)
                same = all(int(ez[f][i]) == int(oez[f]) for f in ("max", "max_q", "max_t")) and np.array_equal(cigs[i], ocig)
                total += 1
                if not same:
                    bad += 1
                    print("MISMATCH seed %d case %d qlen %d tlen %d zdrop %d flag %#x" % (s, i, len(q), len(t), zd, fl))
    v = stats() - s0
    print("short jobs: %d cases against the oracle's kswcpp at the full band, %d mismatching" % (total, bad))
    print("band of 24: %d jobs tried, %d proved; failed check 1 %d, 2 %d, 3 %d, 4 %d; handed on for another reason %d" % tuple(v[:7]))
    # ---- round 6: the long jobs and the adversarial generator, three scoring schemes
    total2 = bad2 = 0
    s0, l0 = stats(), stats(True)
    for s in range(seeds):
        scoring = SCORINGS[s % len(SCORINGS)]
        P2, op2 = params_for(scoring)
        os.environ["MA_KSW_GRP"] = "1033"
        for every in (True, False):
            if every:
                os.environ["MA_KSW_BAND_ALL"] = "1"
            else:
                os.environ.pop("MA_KSW_BAND_ALL", None)
            cases = (long_extension_cases(700, 8000 + s) + adversarial_cases(1500, 8100 + s, 24, scoring, 120, 254, 1000)[0]
                     + adversarial_cases(300, 8200 + s, 120, scoring, 700, 1600, 1000)[0])
            ez, cigs = ma_amd.ksw_batch(P2, cases, pipeline_semantics=True)
            for i, (q, t, w, zd, fl) in enumerate(cases):
                oez, ocig = or_ksw(op2, q, t, w, zd, fl)
                same = all(int(ez[f][i]) == int(oez[f]) for f in ("max", "max_q", "max_t")) and np.array_equal(cigs[i], ocig)
                total2 += 1
                if not same:
                    bad2 += 1
                    print("MISMATCH (round 6 cases) seed %d case %d qlen %d tlen %d zdrop %d flag %#x scoring %s" % (s, i, len(q), len(t), zd, fl, scoring))
    os.environ.pop("MA_KSW_BAND_ALL", None)
    v, vl = stats() - s0, stats(True) - l0
    print("long and adversarial jobs: %d cases against the oracle's kswcpp at the full band, %d mismatching" % (total2, bad2))
    print("band of 24: %d jobs tried, %d proved; failed check 1 %d, 2 %d, 3 %d, 4 %d; handed on for another reason %d" % tuple(v[:7]))
    print("band of 120: %d jobs tried, %d proved; failed check 1 %d, 2 %d, 3 %d, 4 %d; handed on for another reason %d" % tuple(vl[:7]))
    return 1 if bad or bad2 else 0


if __name__ == "__main__":
    sys.exit(main())
