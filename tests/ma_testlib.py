"""Shared helpers for the test-suite: synthetic genomes/reads, case-file writers, oracle loader.

The oracle (oracle/libma_oracle.so) is TEST INFRASTRUCTURE: it is only ever loaded from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DUMP = os.path.join(ORACLE_DIR, "_ref", "ref_dump")
ORACLE_DUMP = os.path.join(ORACLE_DIR, "oracle_dump")


def have_ref():
    return os.path.exists(REF_DUMP)


def build_oracle():
    if not os.path.exists(os.path.join(ORACLE_DIR, "libma_oracle.so")) or not os.path.exists(ORACLE_DUMP):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


# ---------------------------------------------------------------------------------------------
# synthetic data
# ---------------------------------------------------------------------------------------------
def rand_genome(seed, contig_lens, repeat_unit=0, repeat_copies=0, repeat_div=0.05):
    rng = np.random.default_rng(seed)
    contigs = [rng.integers(0, 4, size=int(l), dtype=np.uint8) for l in contig_lens]
    if repeat_unit and repeat_copies:
        unit = rng.integers(0, 4, size=repeat_unit, dtype=np.uint8)
        for _ in range(repeat_copies):
            c = contigs[int(rng.integers(0, len(contigs)))]
            if len(c) <= repeat_unit:
                continue
            p = int(rng.integers(0, len(c) - repeat_unit))
            cp = unit.copy()
            mut = rng.random(repeat_unit) < repeat_div
            cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
            c[p:p + repeat_unit] = cp
    return contigs


def revcomp(a):
    a = np.asarray(a, dtype=np.uint8)
    out = a[::-1].copy()
    m = out < 4
    out[m] = 3 - out[m]
    return out


def sample_reads(contigs, n, length, seed, sub=0.01, ins=0.0, dele=0.0, n_rate=0.0, random_frac=0.0):
    rng = np.random.default_rng(seed)
    reads = []
    for i in range(n):
        if rng.random() < random_frac:
            reads.append(rng.integers(0, 4, size=length, dtype=np.uint8))
            continue
        c = contigs[int(rng.integers(0, len(contigs)))]
        L = min(length, len(c))
        p = int(rng.integers(0, len(c) - L + 1))
        src = c[p:p + L]
        out = []
        for b in src:
            r = rng.random()
            if r < dele:
                continue
            if r < dele + ins:
                out.append(int(rng.integers(0, 4)))
            if rng.random() < sub:
                b = (int(b) + int(rng.integers(1, 4))) % 4
            if rng.random() < n_rate:
                b = 4
            out.append(int(b))
        rd = np.array(out, dtype=np.uint8)
        if i % 2 == 1:
            rd = revcomp(rd)
        reads.append(rd)
    return reads


def write_case(path, contigs, reads, names=None):
    with open(path, "wb") as f:
        f.write(b"MACASE01")
        f.write(struct.pack("<I", len(contigs)))
        for i, c in enumerate(contigs):
            nm = (names[i] if names else "chr%d" % (i + 1)).encode()
            f.write(struct.pack("<I", len(nm)))
            f.write(nm)
            f.write(struct.pack("<Q", len(c)))
            f.write(np.asarray(c, dtype=np.uint8).tobytes())
        f.write(struct.pack("<I", len(reads)))
        for r in reads:
            f.write(struct.pack("<I", len(r)))
            f.write(np.asarray(r, dtype=np.uint8).tobytes())


def read_case(path):
    with open(path, "rb") as f:
        assert f.read(8) == b"MACASE01"
        (nc,) = struct.unpack("<I", f.read(4))
        contigs, names = [], []
        for _ in range(nc):
            (nl,) = struct.unpack("<I", f.read(4))
            names.append(f.read(nl).decode())
            (ln,) = struct.unpack("<Q", f.read(8))
            contigs.append(np.frombuffer(f.read(ln), dtype=np.uint8).copy())
        (nr,) = struct.unpack("<I", f.read(4))
        reads = []
        for _ in range(nr):
            (ln,) = struct.unpack("<I", f.read(4))
            reads.append(np.frombuffer(f.read(ln), dtype=np.uint8).copy())
    return contigs, reads, names


def write_ksw_cases(path, cases):
    """cases: list of (q, t, w, zdrop, flag)"""
    with open(path, "wb") as f:
        f.write(b"KSWCAS01")
        f.write(struct.pack("<I", len(cases)))
        for q, t, w, zdrop, flag in cases:
            f.write(struct.pack("<iiiii", len(q), len(t), w, zdrop, flag))
            f.write(np.asarray(q, dtype=np.uint8).tobytes())
            f.write(np.asarray(t, dtype=np.uint8).tobytes())


def read_ksw_cases(path):
    out = []
    with open(path, "rb") as f:
        assert f.read(8) == b"KSWCAS01"
        (n,) = struct.unpack("<I", f.read(4))
        for _ in range(n):
            ql, tl, w, zd, fl = struct.unpack("<iiiii", f.read(20))
            q = np.frombuffer(f.read(ql), dtype=np.uint8).copy()
            t = np.frombuffer(f.read(tl), dtype=np.uint8).copy()
            out.append((q, t, w, zd, fl))
    return out


KSW_EXTZ = 0x40
KSW_RIGHT = 0x02
KSW_REV = 0x80


def rand_ksw_cases(n, seed, max_len=200, long_frac=0.0):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        if rng.random() < long_frac:
            ql = int(rng.integers(1200, 1800))
            tl = int(rng.integers(1200, 2600))
        else:
            ql = int(rng.integers(1, max_len))
            tl = int(rng.integers(1, max_len))
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        # derive the query from the target with errors so alignments are non-trivial
        src = t[: min(tl, ql)]
        q = []
        er = rng.choice([0.0, 0.02, 0.1, 0.3])
        for b in src:
            r = rng.random()
            if r < er / 3:
                continue
            if r < 2 * er / 3:
                q.append(int(rng.integers(0, 4)))
            if rng.random() < er / 3:
                b = (int(b) + 1) % 4
            q.append(int(b))
        while len(q) < ql:
            q.append(int(rng.integers(0, 4)))
        q = np.array(q[:ql], dtype=np.uint8)
        if rng.random() < 0.1:
            q[rng.integers(0, ql)] = 4
        if rng.random() < 0.1:
            t[rng.integers(0, tl)] = 4
        mode = int(rng.integers(0, 3))
        if mode == 0:
            flag, zdrop = 0, -1
            w = max(20, abs(tl - ql) + 10) if rng.random() < 0.7 else int(rng.integers(1, 60))
        else:
            flag = KSW_EXTZ if mode == 1 else (KSW_EXTZ | KSW_RIGHT | KSW_REV)
            zdrop = int(rng.choice([200, 50, 10, -1]))
            w = int(rng.choice([512, 64, 20, 5, 100]))
        cases.append((q, t, w, zdrop, flag))
    return cases


# ---------------------------------------------------------------------------------------------
# running the dump tools
# ---------------------------------------------------------------------------------------------
def run_ref(*args):
    subprocess.check_call([REF_DUMP] + [str(a) for a in args], stdout=subprocess.DEVNULL)


def run_oracle(*args):
    build_oracle()
    subprocess.check_call([ORACLE_DUMP] + [str(a) for a in args], stdout=subprocess.DEVNULL)


def first_diff(a_path, b_path, context=3):
    with open(a_path) as fa, open(b_path) as fb:
        la, lb = fa.readlines(), fb.readlines()
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            lo = max(0, i - context)
            return "line %d:\n A: %s B: %s context A:\n%s" % (i + 1, x, y, "".join(la[lo:i + 2]))
    if len(la) != len(lb):
        return "length differs: %d vs %d" % (len(la), len(lb))
    return None
