"""ctypes binding of include/ma_amd.h (the drop-in C ABI)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # MA_AMD_LIB: an alternative build of the same library (e.g. the -DMA_KSW_PROF diagnostics build)
    return os.environ.get("MA_AMD_LIB") or os.path.join(_HERE, "libma_amd.so")


class MaError(RuntimeError):
    pass


class Params(C.Structure):
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
        ("search_inversions", C.c_int32), ("zdrop_inversion", C.c_int32), ("use_paired_reads", C.c_int32), ("libm_probe", C.c_int32),
        ("mean_paired_dist", C.c_double), ("std_paired_dist", C.c_double), ("paired_bonus", C.c_double),
    ]

    @staticmethod
    def preset(name="default"):
        """ParameterSetManager::setSelected (parameter.h:1163-1170): default, illumina, illuminapaired, pacbio, nanopore."""
        p = Params()
        _chk(lib().ma_params_preset(name.encode(), C.byref(p)))
        return p


SEGMENT_DT = np.dtype([("q_start", "<i8"), ("q_size", "<i8"), ("sa_start", "<i8"), ("sa_start_rc", "<i8"),
                       ("sa_size", "<i8")])
SEED_DT = np.dtype([("q_start", "<i8"), ("len", "<i8"), ("r_start", "<i8"), ("delta", "<i8"), ("ambiguity", "<u4"),
                    ("on_forward", "<u4")])
EZ_DT = np.dtype([(k, "<i4") for k in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score",
                                       "reach_end", "n_cigar")])
ALIGNMENT_DT = np.dtype([("begin_ref", "<i8"), ("end_ref", "<i8"), ("begin_q", "<i8"), ("end_q", "<i8"),
                         ("score", "<i8"), ("soc_index", "<u4"), ("n_ops", "<u4"), ("ops_off", "<u8"),
                         ("secondary", "<u4"), ("supplementary", "<u4"), ("mapq", "<f8")])
SOC_DT = np.dtype([("acc_len", "<u8"), ("ambiguity", "<u4"), ("n_seeds", "<u4"), ("begin", "<u4"), ("end", "<u4")])
KSW_JOB_DT = np.dtype([("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"), ("zdrop", "<i4"), ("flag", "<i4"),
                       ("reserved", "<u4"), ("q_off", "<u8"), ("t_off", "<u8")])

_lib = None


def lib():
    """Loads libma_amd.so; fails loudly when the HIP extension was not built."""
    global _lib
    if _lib is None:
        try:
            # PyTorch-ROCm bundles its own HIP runtime; load it first so that libma_amd.so binds to the same
            # one (two runtimes in one process cannot both own the device)
            import torch  # noqa: F401
        except ImportError:
            pass
        p = lib_path()
        if not os.path.exists(p):
            raise MaError("libma_amd.so is missing (%s): run __graft_entry__.build(); there is no CPU fallback" % p)
        _lib = C.CDLL(p)
        _lib.ma_last_error.restype = C.c_char_p
    return _lib


def _chk(rc):
    if rc != 0:
        raise MaError(lib().ma_last_error().decode(errors="replace"))


def _ptr(a, t=C.c_void_p):
    if a is None:
        return None
    return a.ctypes.data_as(t)


def device_count():
    n = C.c_int(0)
    _chk(lib().ma_device_count(C.byref(n)))
    return n.value


def set_device(d):
    _chk(lib().ma_set_device(C.c_int(d)))


def bind_host_thread(device, mode=0):
    """Pins the calling thread (and the threads it starts later) to the CPUs next to GPU `device` (ma_host_bind_thread); mode 1:
    to the other CPUs, -1: to all again.  Returns the CPUs of the new mask, 0 when it was left alone."""
    n = C.c_int(0)
    _chk(lib().ma_host_bind_thread(C.c_int(device), C.c_int(mode), C.byref(n)))
    return n.value


class HostArray:
    """Page-locked host memory (ma_host_alloc) as a numpy array: uploads from it and downloads into it are DMA transfers
    that run asynchronously on the batch's stream, so the copies of one batch hide behind the kernels of another."""

    def __init__(self, n, dtype):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self.p = C.c_void_p()
        nbytes = max(self.n * self.dtype.itemsize, 64)
        _chk(lib().ma_host_alloc(C.c_uint64(nbytes), C.byref(self.p)))
        buf = (C.c_char * nbytes).from_address(self.p.value)
        self.a = np.frombuffer(buf, dtype=self.dtype, count=self.n)

    @property
    def ptr(self):
        return self.p.value

    def close(self):
        if self.p:
            self.a = None
            lib().ma_host_free(self.p)
            self.p = None


class Index:
    """Device-resident FMD-index + pack (ma_index*)."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def from_arrays(bwt, sa, L2, primary, ref_len, pac, contig_starts, contig_lens):
        bwt = np.ascontiguousarray(bwt, dtype=np.uint32)
        sa = np.ascontiguousarray(sa, dtype=np.int64)
        L2 = np.ascontiguousarray(L2, dtype=np.uint64)
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        cs = np.ascontiguousarray(contig_starts, dtype=np.uint64)
        cl = np.ascontiguousarray(contig_lens, dtype=np.uint64)
        h = C.c_void_p()
        _chk(lib().ma_index_create(_ptr(bwt), C.c_uint64(len(bwt)), _ptr(sa), C.c_uint64(len(sa)), _ptr(L2),
                                   C.c_int64(int(primary)), C.c_uint64(int(ref_len)), _ptr(pac), C.c_int32(len(cs)),
                                   _ptr(cs), _ptr(cl), C.byref(h)))
        return Index(h)

    @staticmethod
    def build(contigs):
        """GPU index construction from N-free contigs (list of uint8 code arrays)."""
        lens = np.array([len(c) for c in contigs], dtype=np.uint64)
        cat = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint8) for c in contigs]))
        h = C.c_void_p()
        _chk(lib().ma_index_build(C.c_int32(len(lens)), _ptr(lens), _ptr(cat), C.byref(h)))
        return Index(h)

    @staticmethod
    def build_device(contig_lens, d_codes_ptr):
        lens = np.ascontiguousarray(contig_lens, dtype=np.uint64)
        h = C.c_void_p()
        _chk(lib().ma_index_build_device(C.c_int32(len(lens)), _ptr(lens), C.c_void_p(int(d_codes_ptr)), C.byref(h)))
        return Index(h)

    def sizes(self):
        nw, ns, rl, nc = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        _chk(lib().ma_index_sizes(self.h, C.byref(nw), C.byref(ns), C.byref(rl), C.byref(nc)))
        return nw.value, ns.value, rl.value, nc.value

    def download(self):
        nw, ns, rl, nc = self.sizes()
        bwt = np.empty(nw, dtype=np.uint32)
        sa = np.empty(ns, dtype=np.int64)
        L2 = np.zeros(5, dtype=np.uint64)
        primary = C.c_int64()
        pac = np.empty((rl // 2 + 3) // 4, dtype=np.uint8)
        cs = np.empty(nc, dtype=np.uint64)
        cl = np.empty(nc, dtype=np.uint64)
        _chk(lib().ma_index_download(self.h, _ptr(bwt), _ptr(sa), _ptr(L2), C.byref(primary), _ptr(pac), _ptr(cs),
                                     _ptr(cl)))
        return dict(bwt=bwt, sa=sa, L2=L2, primary=primary.value, ref_len=rl, pac=pac, contig_starts=cs,
                    contig_lens=cl)

    def store(self, prefix, names=None, title=None):
        """Write the index as the reference's <prefix>.bwt/.sa/.pac/.ann/.amb files (FMIndex::vStoreFMIndex
        fMIndex.h:515-549, Pack::vStoreCollection pack.h:230-269,725-770; same layout as storeIndex of the C++ host
        layer): maCMD / FMIndex(prefix) load an index that was built on the GPU."""
        if title:  # the genome file `maCMD -x` takes (execution-context.h:60-136)
            import json as _json
            import os as _os
            with open(_os.path.join(_os.path.dirname(prefix), title + ".json"), "w") as f:
                f.write(_json.dumps({"name": title, "prefix": _os.path.basename(prefix), "type": "MA Genome",
                                     "version": {"major": 1, "minor": 0}}, indent=4, sort_keys=True) + "\n")
        d = self.download()
        n = int(d["ref_len"])
        F = n // 2
        with open(prefix + ".bwt", "wb") as f:
            f.write(np.int64(d["primary"]).tobytes())
            f.write(np.ascontiguousarray(d["L2"][1:5], dtype=np.uint64).tobytes())
            d["bwt"].tofile(f)
        with open(prefix + ".sa", "wb") as f:
            f.write(np.int64(d["primary"]).tobytes())
            f.write(np.ascontiguousarray(d["L2"][1:5], dtype=np.uint64).tobytes())
            f.write(np.int32(32).tobytes())
            f.write(np.uint64(n).tobytes())
            d["sa"][1:].tofile(f)
        with open(prefix + ".pac", "wb") as f:
            d["pac"][:(F + 3) // 4].tofile(f)
            if F % 4 == 0:
                f.write(b"\0")
            f.write(bytes([F % 4]))
        nc = len(d["contig_starts"])
        with open(prefix + ".ann", "w") as f:
            f.write("%d %d 0\n" % (F, nc))
            for i in range(nc):
                f.write("0 %s none\n%d %d 0\n" % (names[i] if names else "chr%d" % (i + 1), int(d["contig_starts"][i]),
                                                  int(d["contig_lens"][i])))
        with open(prefix + ".amb", "w") as f:
            f.write("%d %d 0\n" % (F, nc))

    def extract(self, begin, end):
        """Pack::vExtract for ranges [begin[i], end[i]) of the doubled text: list of uint8 code arrays."""
        b = np.ascontiguousarray(begin, dtype=np.uint64)
        e = np.ascontiguousarray(end, dtype=np.uint64)
        n = np.where(e > b, e - b, 0)
        out = np.empty(int(n.sum()) + 1, dtype=np.uint8)
        _chk(lib().ma_pack_extract(self.h, _ptr(b), _ptr(e), C.c_uint64(len(b)), _ptr(out)))
        cuts = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        return [out[cuts[i]:cuts[i + 1]] for i in range(len(b))]

    def extend_backward(self, ik, c):
        ik = np.ascontiguousarray(ik, dtype=np.int64).reshape(-1, 3)
        c = np.ascontiguousarray(c, dtype=np.uint8)
        ok = np.empty_like(ik)
        _chk(lib().ma_extend_backward_batch(self.h, _ptr(ik), _ptr(c), C.c_uint64(len(c)), _ptr(ok)))
        return ok

    def bwt_sa(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        pos = np.empty_like(rows)
        _chk(lib().ma_bwt_sa_batch(self.h, _ptr(rows), C.c_uint64(len(rows)), _ptr(pos)))
        return pos

    def close(self):
        if self.h:
            lib().ma_index_destroy(self.h)
            self.h = None


def ksw_batch(params, cases, cigar_cap=None, pipeline_semantics=False):
    """cases: list of (q, t, w, zdrop, flag) -> (ez structured array, list of cigar arrays).
    pipeline_semantics: ma_ksw_ext_batch (only max, max_q, max_t, cigar defined)."""
    n = len(cases)
    jobs = np.zeros(n, dtype=KSW_JOB_DT)
    qs, ts = [], []
    qo = to = 0
    for i, (q, t, w, zd, fl) in enumerate(cases):
        jobs[i] = (len(q), len(t), w, zd, fl, 0, qo, to)
        qs.append(np.asarray(q, dtype=np.uint8))
        ts.append(np.asarray(t, dtype=np.uint8))
        qo += len(q)
        to += len(t)
    qb = np.ascontiguousarray(np.concatenate(qs + [np.zeros(1, dtype=np.uint8)]))
    tb = np.ascontiguousarray(np.concatenate(ts + [np.zeros(1, dtype=np.uint8)]))
    if cigar_cap is None:
        cigar_cap = qo + to + 2 * n + 16
    ez = np.zeros(n, dtype=EZ_DT)
    off = np.zeros(n + 1, dtype=np.uint64)
    cig = np.zeros(cigar_cap, dtype=np.uint32)
    fn = lib().ma_ksw_ext_batch if pipeline_semantics else lib().ma_ksw_batch
    _chk(fn(C.byref(params), _ptr(jobs), C.c_uint64(n), _ptr(qb), C.c_uint64(len(qb)), _ptr(tb),
            C.c_uint64(len(tb)), _ptr(ez), _ptr(off), _ptr(cig), C.c_uint64(cigar_cap)))
    cigs = [cig[int(off[i]):int(off[i]) + int(ez["n_cigar"][i])].copy() for i in range(n)]
    return ez, cigs


class Batch:
    """Device-resident batch of reads and stage outputs (ma_batch*)."""

    def __init__(self, index, params, max_reads, max_bases):
        self.index = index
        self.params = params
        self.h = C.c_void_p()
        _chk(lib().ma_batch_create(index.h, C.byref(params), C.c_uint64(max_reads), C.c_uint64(max_bases),
                                   C.byref(self.h)))
        self.n = 0

    def set_stream(self, stream_ptr):
        _chk(lib().ma_batch_set_stream(self.h, C.c_void_p(int(stream_ptr))))

    def set_blocking_sync(self, on=True):
        """waits of this batch sleep on an interrupt-driven event instead of spinning (hosts with more waiting threads than cores)"""
        _chk(lib().ma_batch_set_blocking_sync(self.h, C.c_int(1 if on else 0)))

    def enable_timing(self, on=True):
        _chk(lib().ma_batch_enable_timing(self.h, C.c_int(1 if on else 0)))

    def set_reads(self, reads):
        """reads: list of uint8 code arrays"""
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        if len(reads):
            off[1:] = np.cumsum([len(r) for r in reads])
            cat = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint8) for r in reads]
                                                      + [np.zeros(1, dtype=np.uint8)]))
        else:
            cat = np.zeros(1, dtype=np.uint8)
        _chk(lib().ma_batch_set_reads(self.h, _ptr(cat), _ptr(off), C.c_uint64(len(reads))))
        self.n = len(reads)

    def set_reads_flat(self, codes_ptr, offsets_ptr, n_reads):
        """reads that already are one host array of codes + CSR offsets (n + 1 u64, starting at 0), e.g. in page-locked memory"""
        _chk(lib().ma_batch_set_reads(self.h, C.c_void_p(int(codes_ptr)), C.c_void_p(int(offsets_ptr)), C.c_uint64(n_reads)))
        self.n = n_reads

    def set_reads_device(self, d_codes_ptr, d_offsets_ptr, n_reads, n_bases):
        _chk(lib().ma_batch_set_reads_device(self.h, C.c_void_p(int(d_codes_ptr)), C.c_void_p(int(d_offsets_ptr)),
                                             C.c_uint64(n_reads), C.c_uint64(n_bases)))
        self.n = n_reads

    def seed(self):
        _chk(lib().ma_seed_batch(self.h))

    def extract(self):
        _chk(lib().ma_extract_seeds_batch(self.h))

    def chain(self):
        _chk(lib().ma_chain_batch(self.h))

    def dp(self):
        _chk(lib().ma_dp_batch(self.h))

    def align(self):
        _chk(lib().ma_align_batch(self.h))

    def sync(self):
        _chk(lib().ma_batch_sync(self.h))

    def counts(self):
        v = [C.c_uint64() for _ in range(7)]
        _chk(lib().ma_batch_counts(self.h, *[C.byref(x) for x in v]))
        keys = ("segments", "seeds", "hsets", "hseeds", "alignments", "ops_cap", "aligned_reads")
        return dict(zip(keys, [x.value for x in v]))

    def counters(self):
        out = np.zeros(8, dtype=np.uint64)
        _chk(lib().ma_batch_counters(self.h, _ptr(out)))
        return out

    def dp_jobs(self):
        """(n, 8) int32: qlen, tlen, w, zdrop, flag, zdropped, max_q, max_t of every kswcpp call of the last dp stage."""
        n = C.c_uint64()
        _chk(lib().ma_batch_get_dp_jobs(self.h, C.byref(n), None, C.c_uint64(0)))
        out = np.zeros((int(n.value), 8), dtype=np.int32)
        if n.value:
            _chk(lib().ma_batch_get_dp_jobs(self.h, C.byref(n), _ptr(out), C.c_uint64(n.value)))
        return out

    def kernel_ms(self):
        out = np.zeros(8, dtype=np.float32)
        _chk(lib().ma_batch_kernel_ms(self.h, _ptr(out)))
        return out

    def segments(self):
        c = self.counts()
        off = np.zeros(self.n + 1, dtype=np.uint64)
        segs = np.zeros(c["segments"], dtype=SEGMENT_DT)
        _chk(lib().ma_batch_get_segments(self.h, _ptr(off), _ptr(segs)))
        return off, segs

    def seeds(self):
        c = self.counts()
        off = np.zeros(self.n + 1, dtype=np.uint64)
        seeds = np.zeros(c["seeds"], dtype=SEED_DT)
        _chk(lib().ma_batch_get_seeds(self.h, _ptr(off), _ptr(seeds)))
        return off, seeds

    def hsets(self):
        c = self.counts()
        hoff = np.zeros(self.n + 1, dtype=np.uint64)
        soff = np.zeros(c["hsets"] + 1, dtype=np.uint64)
        soc = np.zeros(c["hsets"], dtype=np.uint32)
        seeds = np.zeros(c["hseeds"], dtype=SEED_DT)
        _chk(lib().ma_batch_get_hsets(self.h, _ptr(hoff), _ptr(soff), _ptr(soc), _ptr(seeds)))
        return hoff, soff, soc, seeds[: int(soff[-1])]

    def set_seeds(self, off, seeds):
        """Extracted seeds from elsewhere (CSR per read) -> chain()."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        seeds = np.ascontiguousarray(np.concatenate([seeds, np.zeros(1, dtype=SEED_DT)]))
        _chk(lib().ma_batch_set_seeds(self.h, _ptr(off), _ptr(seeds)))

    def socs(self, heap=False):
        """The SoC queue of every read: strips in pop order, or (heap=True) the queue's array as the sweep leaves it;
        returns (soc_off, socs, seed_off, seeds re-sorted by reference position)."""
        fn = lib().ma_batch_get_soc_heap if heap else lib().ma_batch_get_socs
        n = C.c_uint64()
        _chk(fn(self.h, C.byref(n), None, None, None, None))
        c = self.counts()
        soff = np.zeros(self.n + 1, dtype=np.uint64)
        socs = np.zeros(int(n.value) + 1, dtype=SOC_DT)
        doff = np.zeros(self.n + 1, dtype=np.uint64)
        seeds = np.zeros(c["seeds"] + 1, dtype=SEED_DT)
        _chk(fn(self.h, C.byref(n), _ptr(soff), _ptr(socs), _ptr(doff), _ptr(seeds)))
        return soff, socs[: int(n.value)], doff, seeds[: c["seeds"]]

    def set_soc_heap(self, soc_off, socs, seed_off, seeds):
        """SoC queues swept elsewhere (layout of socs(heap=True)) -> chain() only harmonizes."""
        soc_off = np.ascontiguousarray(soc_off, dtype=np.uint64)
        seed_off = np.ascontiguousarray(seed_off, dtype=np.uint64)
        socs = np.ascontiguousarray(np.concatenate([socs, np.zeros(1, dtype=SOC_DT)]))
        seeds = np.ascontiguousarray(np.concatenate([seeds, np.zeros(1, dtype=SEED_DT)]))
        _chk(lib().ma_batch_set_soc_heap(self.h, _ptr(soc_off), _ptr(socs), _ptr(seed_off), _ptr(seeds)))

    def set_alignments(self, off, alns, ops):
        """Alignments computed elsewhere (NeedlemanWunsch order) -> the MappingQuality kernel alone."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        alns = np.ascontiguousarray(np.concatenate([alns, np.zeros(1, dtype=ALIGNMENT_DT)]))
        ops = np.ascontiguousarray(np.concatenate([np.asarray(ops, dtype=np.uint64), np.zeros(2, dtype=np.uint64)]))
        _chk(lib().ma_batch_set_alignments(self.h, _ptr(off), _ptr(alns), _ptr(ops)))

    def _alns(self, fn):
        c = self.counts()
        off = np.zeros(self.n + 1, dtype=np.uint64)
        alns = np.zeros(c["alignments"], dtype=ALIGNMENT_DT)
        ops = np.zeros(2 * c["ops_cap"] + 2, dtype=np.uint64)
        _chk(fn(self.h, _ptr(off), _ptr(alns), _ptr(ops)))
        return off, alns[: int(off[-1])], ops

    def alignments(self):
        return self._alns(lib().ma_batch_get_alignments)

    def mapq_alignments(self):
        return self._alns(lib().ma_batch_get_mapq_alignments)

    def mapq_alignments_into(self, off, alns, ops):
        """The MappingQuality records into caller-owned host arrays (HostArray: u64[n + 1], ALIGNMENT_DT[>= alignments],
        u64[>= 2 * ops_cap + 2]); returns None when they are too small (the caller grows them), else the counts."""
        c = self.counts()
        if off.n < self.n + 1 or alns.n < c["alignments"] + 1 or ops.n < 2 * c["ops_cap"] + 2:
            return None
        _chk(lib().ma_batch_get_mapq_alignments(self.h, C.c_void_p(off.ptr), C.c_void_p(alns.ptr), C.c_void_p(ops.ptr)))
        return c

    # ---- double-buffered I/O (ma_batch_stage_reads ...): upload of the next reads and download of the last results beside the kernels
    def stage_reads_flat(self, codes_ptr, offsets_ptr, n_reads):
        _chk(lib().ma_batch_stage_reads(self.h, C.c_void_p(int(codes_ptr)), C.c_void_p(int(offsets_ptr)), C.c_uint64(n_reads)))
        self._staged_n = n_reads

    def use_staged_reads(self):
        _chk(lib().ma_batch_use_staged_reads(self.h))
        self.n = self._staged_n

    def start_mapq_download(self, off, alns, ops):
        """mapq_alignments_into without the wait: returns None when the arrays are too small, else the counts; the arrays are
        complete after finish_download()."""
        c = self.counts()
        if off.n < self.n + 1 or alns.n < c["alignments"] + 1 or ops.n < 2 * c["ops_cap"] + 2:
            return None
        _chk(lib().ma_batch_start_mapq_download(self.h, C.c_void_p(off.ptr), C.c_void_p(alns.ptr), C.c_void_p(ops.ptr)))
        return c

    def finish_download(self):
        _chk(lib().ma_batch_finish_download(self.h))

    def close(self):
        if self.h:
            lib().ma_batch_destroy(self.h)
            self.h = None
