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


def _mutate(src, rng, sub):
    out = np.array(src, dtype=np.uint8).copy()
    hit = rng.random(len(out)) < sub
    out[hit] = (out[hit] + rng.integers(1, 4, size=int(hit.sum()))) % 4
    return out


def sample_inversion_reads(contigs, n, length, seed, inv_min=120, inv_max=400, sub=0.01):
    """Reads whose inner stretch is reverse-complemented in place (a small inversion without seeds of its own
    orientation in the chain): what SmallInversions (smallInversions.h) looks for."""
    rng = np.random.default_rng(seed)
    reads = []
    for i in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        L = min(length, len(c))
        p = int(rng.integers(0, len(c) - L + 1))
        rd = _mutate(c[p:p + L], rng, sub)
        w = int(rng.integers(inv_min, inv_max + 1))
        a = int(rng.integers(L // 4, max(L // 4 + 1, 3 * L // 4 - w)))
        rd[a:a + w] = revcomp(rd[a:a + w])
        if i % 2 == 1:
            rd = revcomp(rd)
        reads.append(rd)
    return reads


def sample_pairs(contigs, n, length, seed, insert_mean=400, insert_std=60, sub=0.01, far_frac=0.1, same_strand_frac=0.05,
                 random_mate_frac=0.08):
    """Mate pairs (reads 2k, 2k+1) as PairedReads (pairedReads.cpp) expects them after PairedFileReader: opposite
    strands, outer distance ~ insert_mean; some pairs too far apart, on the same strand, or with a random mate."""
    rng = np.random.default_rng(seed)
    reads = []
    for k in range(n):
        c = contigs[int(rng.integers(0, len(contigs)))]
        d = max(length + 1, int(rng.normal(insert_mean, insert_std)))
        if rng.random() < far_frac:
            d = int(rng.integers(2000, 6000))
        d = min(d, len(c) - 1)
        p = int(rng.integers(0, len(c) - d))
        m1 = _mutate(c[p:p + length], rng, sub)
        m2 = _mutate(c[p + d - length:p + d], rng, sub)
        if rng.random() >= same_strand_frac:
            m2 = revcomp(m2)
        if k % 2 == 1:  # the pair seen from the other strand
            m1, m2 = revcomp(m1), revcomp(m2)
            m1, m2 = m2, m1
        r = rng.random()
        if r < random_mate_frac / 2:
            m1 = rng.integers(0, 4, size=length, dtype=np.uint8)
        elif r < random_mate_frac:
            m2 = rng.integers(0, 4, size=length, dtype=np.uint8)
        elif r < random_mate_frac * 1.25:
            m1 = rng.integers(0, 4, size=length, dtype=np.uint8)
            m2 = rng.integers(0, 4, size=length, dtype=np.uint8)
        reads += [m1, m2]
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


# ---------------------------------------------------------------------------------------------
# dump parsing (the text format written by ref_dump / oracle_dump / host_emul)
# ---------------------------------------------------------------------------------------------
def _open(path):
    import gzip
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path)


def parse_pipe_dump(path):
    reads = []
    cur = None
    with _open(path) as f:
        for line in f:
            t = line.split()
            k = t[0]
            if k == "R":
                cur = dict(len=int(t[2]), segs=[], seeds=[], socs=[], hsets=[], alns=[], mq=[])
                reads.append(cur)
            elif k == "s":
                cur["segs"].append(tuple(int(x) for x in t[1:6]))
            elif k == "d":
                cur["seeds"].append(tuple(int(x) for x in t[1:7]))
            elif k == "c":
                cur["socs"].append(dict(idx=int(t[1]), score=int(t[2]), amb=int(t[3]), seeds=[]))
            elif k == "e":
                cur["socs"][-1]["seeds"].append(tuple(int(x) for x in t[1:7]))
            elif k == "h":
                cur["hsets"].append(dict(soc=int(t[1]), seeds=[]))
            elif k == "g":
                cur["hsets"][-1]["seeds"].append(tuple(int(x) for x in t[1:7]))
            elif k == "a":
                ops = [tuple(int(y) for y in x.split(":")) for x in t[8:]]
                cur["alns"].append(dict(bref=int(t[1]), eref=int(t[2]), bq=int(t[3]), eq=int(t[4]), score=int(t[5]),
                                        soc=int(t[6]), ops=ops))
            elif k == "m":
                cur["mq"].append(dict(bref=int(t[1]), eref=int(t[2]), bq=int(t[3]), eq=int(t[4]), score=int(t[5]),
                                      secondary=int(t[6]), supplementary=int(t[7]), mapq=float(t[8])))
    return reads


def parse_ksw_dump(path):
    out = []
    with _open(path) as f:
        for line in f:
            t = line.split()
            v = [int(x) for x in t[2:13]]
            out.append(dict(max=v[0], zdropped=v[1], max_q=v[2], max_t=v[3], mqe=v[4], mqe_t=v[5], mte=v[6],
                            mte_q=v[7], score=v[8], reach_end=v[9], n_cigar=v[10],
                            cigar=[int(x) for x in t[13:]]))
    return out


def gunzip_to(src_gz, dst):
    import gzip
    import shutil
    with gzip.open(src_gz, "rb") as f, open(dst, "wb") as g:
        shutil.copyfileobj(f, g)
    return dst


# ---------------------------------------------------------------------------------------------
# ctypes binding of the oracle (TEST INFRASTRUCTURE)
# ---------------------------------------------------------------------------------------------
class OrParams(C.Structure):
    _fields_ = [
        ("seeding_technique", C.c_int32), ("min_seed_len", C.c_int32), ("min_ambiguity", C.c_int32),
        ("max_ambiguity", C.c_int32), ("min_seed_size_drop", C.c_int32), ("max_num_soc", C.c_int32),
        ("min_num_soc", C.c_int32), ("harm_score_min", C.c_int32), ("max_score_lookahead", C.c_int32),
        ("switch_qlen", C.c_int32), ("min_delta_dist", C.c_int32), ("max_gap_area", C.c_int32),
        ("padding", C.c_int32), ("bandwidth_ext", C.c_int32), ("min_bandwidth_gap", C.c_int32),
        ("zdrop", C.c_int32), ("sv_penalty", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32),
        ("gap", C.c_int32), ("extend", C.c_int32), ("gap2", C.c_int32), ("extend2", C.c_int32),
        ("disable_heuristics", C.c_int32), ("soc_width", C.c_int32), ("srand_seed", C.c_uint32),
        ("genome_size_disable", C.c_uint64), ("rel_min_seed_size_amount", C.c_double),
        ("harm_score_min_rel", C.c_double), ("soc_score_decrease_tol", C.c_double),
        ("score_diff_tol", C.c_double), ("max_delta_dist", C.c_double), ("min_alignment_score", C.c_int32),
        ("report_n_best", C.c_int32), ("max_supplementary", C.c_int32), ("max_overlap_supplementary", C.c_double),
        ("search_inversions", C.c_int32), ("zdrop_inversion", C.c_int32), ("use_paired_reads", C.c_int32), ("pad_", C.c_int32),
        ("mean_paired_dist", C.c_double), ("std_paired_dist", C.c_double), ("paired_bonus", C.c_double),
    ]


OR_SEGMENT_DT = np.dtype([("q_start", "<i8"), ("q_size", "<i8"), ("sa_start", "<i8"), ("sa_start_rc", "<i8"),
                          ("sa_size", "<i8")])
OR_SEED_DT = np.dtype([("q_start", "<i8"), ("len", "<i8"), ("r_start", "<i8"), ("delta", "<i8"), ("ambiguity", "<u4"),
                       ("on_forward", "<u4")])
OR_EZ_DT = np.dtype([(k, "<i4") for k in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score",
                                          "reach_end", "n_cigar")])
OR_ALN_DT = np.dtype([("begin_ref", "<i8"), ("end_ref", "<i8"), ("begin_q", "<i8"), ("end_q", "<i8"), ("score", "<i8"),
                      ("soc_index", "<u4"), ("n_ops", "<u4"), ("ops_off", "<u8"), ("secondary", "<u4"),
                      ("supplementary", "<u4"), ("mapq", "<f8")])

_orlib = None


def orlib():
    global _orlib
    if _orlib is None:
        build_oracle()
        L = C.CDLL(os.path.join(ORACLE_DIR, "libma_oracle.so"))
        L.ma_or_index_build.restype = C.c_void_p
        L.ma_or_index_from_parts.restype = C.c_void_p
        L.ma_or_align_batch.restype = C.c_void_p
        for fn in ("ma_or_index_n_words", "ma_or_index_n_sa", "ma_or_res_n_ops", "ma_or_res_n_aligned"):
            getattr(L, fn).restype = C.c_uint64
        for fn in ("ma_or_index_bwt", "ma_or_index_sa", "ma_or_index_pac", "ma_or_res_seg_off", "ma_or_res_segs",
                   "ma_or_res_seed_off", "ma_or_res_seeds", "ma_or_res_hset_off", "ma_or_res_hseed_off",
                   "ma_or_res_hset_soc", "ma_or_res_hseeds", "ma_or_res_aln_off", "ma_or_res_alns", "ma_or_res_ops",
                   "ma_or_res_mq_off", "ma_or_res_mq"):
            getattr(L, fn).restype = C.c_void_p
        L.ma_or_bwt_sa.restype = C.c_int64
        _orlib = L
    return _orlib


def or_params(preset="default", seed=1):
    p = OrParams()
    if preset.startswith("illumina"):
        orlib().ma_or_params_illumina(C.byref(p))
    elif preset.startswith("pacbio"):
        orlib().ma_or_params_pacbio(C.byref(p))
    elif preset.startswith("nanopore"):
        orlib().ma_or_params_nanopore(C.byref(p))
    else:
        orlib().ma_or_params_default(C.byref(p))
    if preset.endswith("+mems"):  # the MEMs seeding technique (binarySeeding.h:460-537), selected by no preset
        p.seeding_technique = 2
    p.srand_seed = seed
    return p


def _np_from(ptr, n, dt):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt).copy()


class OrIndex:
    def __init__(self, h, contig_lens):
        self.h = C.c_void_p(h)
        self.contig_lens = np.asarray(contig_lens, dtype=np.uint64)
        self.contig_starts = np.concatenate([[0], np.cumsum(self.contig_lens)[:-1]]).astype(np.uint64)

    @staticmethod
    def build(contigs):
        lens = np.array([len(c) for c in contigs], dtype=np.uint64)
        cat = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint8) for c in contigs]))
        h = orlib().ma_or_index_build(C.c_int32(len(lens)), lens.ctypes.data_as(C.c_void_p),
                                      cat.ctypes.data_as(C.c_void_p))
        return OrIndex(h, lens)

    @staticmethod
    def from_parts(d):
        bwt = np.ascontiguousarray(d["bwt"], dtype=np.uint32)
        sa = np.ascontiguousarray(d["sa"], dtype=np.int64)
        L2 = np.ascontiguousarray(d["L2"], dtype=np.uint64)
        pac = np.ascontiguousarray(d["pac"], dtype=np.uint8)
        cs = np.ascontiguousarray(d["contig_starts"], dtype=np.uint64)
        cl = np.ascontiguousarray(d["contig_lens"], dtype=np.uint64)
        h = orlib().ma_or_index_from_parts(bwt.ctypes.data_as(C.c_void_p), C.c_uint64(len(bwt)),
                                           sa.ctypes.data_as(C.c_void_p), C.c_uint64(len(sa)),
                                           L2.ctypes.data_as(C.c_void_p), C.c_int64(int(d["primary"])),
                                           C.c_uint64(int(d["ref_len"])), pac.ctypes.data_as(C.c_void_p),
                                           C.c_int32(len(cs)), cs.ctypes.data_as(C.c_void_p),
                                           cl.ctypes.data_as(C.c_void_p))
        return OrIndex(h, cl)

    def arrays(self):
        L = orlib()
        nw = L.ma_or_index_n_words(self.h)
        ns = L.ma_or_index_n_sa(self.h)
        L2 = np.zeros(5, dtype=np.uint64)
        primary = C.c_int64()
        ref_len = C.c_uint64()
        L.ma_or_index_meta(self.h, L2.ctypes.data_as(C.c_void_p), C.byref(primary), C.byref(ref_len))
        F = ref_len.value // 2
        return dict(bwt=_np_from(L.ma_or_index_bwt(self.h), nw, np.uint32),
                    sa=_np_from(L.ma_or_index_sa(self.h), ns, np.int64), L2=L2, primary=primary.value,
                    ref_len=ref_len.value, pac=_np_from(L.ma_or_index_pac(self.h), (F + 3) // 4, np.uint8),
                    contig_starts=self.contig_starts, contig_lens=self.contig_lens)

    def extend_backward(self, ik, c):
        ik = np.ascontiguousarray(ik, dtype=np.int64).reshape(-1, 3)
        ok = np.zeros_like(ik)
        for i in range(len(ik)):
            a = ik[i].copy()
            o = np.zeros(3, dtype=np.int64)
            orlib().ma_or_extend_backward(self.h, a.ctypes.data_as(C.c_void_p), C.c_uint8(int(c[i])),
                                          o.ctypes.data_as(C.c_void_p))
            ok[i] = o
        return ok

    def bwt_sa(self, rows):
        return np.array([orlib().ma_or_bwt_sa(self.h, C.c_int64(int(r))) for r in rows], dtype=np.int64)

    def align(self, reads, params, threads=1):
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in reads])
        cat = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint8) for r in reads]
                                                  + [np.zeros(1, dtype=np.uint8)]))
        L = orlib()
        r = C.c_void_p(L.ma_or_align_batch(self.h, C.byref(params), cat.ctypes.data_as(C.c_void_p),
                                           off.ctypes.data_as(C.c_void_p), C.c_uint64(len(reads)), C.c_int32(threads)))
        n = len(reads)
        res = {}
        res["seg_off"] = _np_from(L.ma_or_res_seg_off(r), n + 1, np.uint64)
        res["segs"] = _np_from(L.ma_or_res_segs(r), int(res["seg_off"][-1]), OR_SEGMENT_DT)
        res["seed_off"] = _np_from(L.ma_or_res_seed_off(r), n + 1, np.uint64)
        res["seeds"] = _np_from(L.ma_or_res_seeds(r), int(res["seed_off"][-1]), OR_SEED_DT)
        res["hset_off"] = _np_from(L.ma_or_res_hset_off(r), n + 1, np.uint64)
        nh = int(res["hset_off"][-1])
        res["hseed_off"] = _np_from(L.ma_or_res_hseed_off(r), nh + 1, np.uint64)
        res["hset_soc"] = _np_from(L.ma_or_res_hset_soc(r), nh, np.uint32)
        res["hseeds"] = _np_from(L.ma_or_res_hseeds(r), int(res["hseed_off"][-1]), OR_SEED_DT)
        res["aln_off"] = _np_from(L.ma_or_res_aln_off(r), n + 1, np.uint64)
        res["alns"] = _np_from(L.ma_or_res_alns(r), int(res["aln_off"][-1]), OR_ALN_DT)
        res["ops"] = _np_from(L.ma_or_res_ops(r), 2 * int(L.ma_or_res_n_ops(r)), np.uint64)
        res["mq_off"] = _np_from(L.ma_or_res_mq_off(r), n + 1, np.uint64)
        res["mq"] = _np_from(L.ma_or_res_mq(r), int(res["mq_off"][-1]), OR_ALN_DT)
        ctr = np.zeros(8, dtype=np.uint64)
        L.ma_or_res_counters(r, ctr.ctypes.data_as(C.c_void_p))
        res["counters"] = ctr
        res["n_aligned"] = int(L.ma_or_res_n_aligned(r))
        L.ma_or_result_free(r)
        return res


def or_ksw(params, q, t, w, zdrop, flag):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    ez = np.zeros(1, dtype=OR_EZ_DT)
    cap = len(q) + len(t) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    n = orlib().ma_or_ksw(C.c_int32(len(q)), q.ctypes.data_as(C.c_void_p), C.c_int32(len(t)),
                          t.ctypes.data_as(C.c_void_p), C.c_int32(w), C.c_int32(zdrop), C.c_int32(flag),
                          C.byref(params), ez.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p), C.c_int32(cap))
    return ez[0], cig[:max(n, 0)].copy()
