"""Round 4: the index pinned at GRCh38 scale (text positions above 2^32, 35-bit SMEM entries, 64-bit row arithmetic, the
bucketed index build at 6.2 Gnt), configs[0] (C1) at its stated genome size against the compiled reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ma_testlib import parse_pipe_dump, write_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616,
          64444167, 46709983, 50818468, 156040895, 57227415]


@pytest.fixture(scope="module")
def gpu_device():
    import ma_amd
    if ma_amd.device_count() < 1:
        pytest.skip("no HIP device")
    ma_amd.set_device(0)
    return 0


def _synth_reads(L, idx, seed, n, rl, sub=0.0, ins=0.0, dele=0.0):
    import torch
    cap = int(n * (rl * (1 + 2 * ins) + 8)) + 1024
    codes = torch.empty(cap, dtype=torch.uint8, device="cuda")
    offs = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    nb = C.c_uint64()
    assert L.ma_synth_reads_device(idx.h, C.c_uint64(seed), C.c_uint64(n), C.c_uint32(rl), C.c_double(sub), C.c_double(ins),
                                   C.c_double(dele), C.c_uint64(0), C.c_void_p(codes.data_ptr()), C.c_void_p(offs.data_ptr()),
                                   C.c_uint64(cap), C.byref(nb)) == 0
    return codes, offs, int(nb.value)


def test_index_pinned_at_grch38_scale(gpu_device):
    """VERDICT r3 item 6.  The index bench.py aligns against (24 contigs of GRCh38's lengths, 3.09 Gnt forward, 6.18 Gnt of
    indexed text, planted repeat families) plus two exact 40-mers planted 100 and 101 times, checked WITHOUT another index
    implementation in the loop:
      (i)   every occ counter of the 64-byte blocks = a recount of the packed BWT symbols before it (fMIndex.cpp:204-264), L2 too;
      (ii)  10^5 random sampled suffix-array rows: T[sa[k]..] < T[sa[k+1]..] for neighbouring samples (rows 32 k, 32 k + 32),
            and the BWT symbol of the row is T[sa[k] - 1] (fMIndex.cpp:266-314, fMIndex.h:788-814);
      (iii) error-free 150 bp and 10 kb reads map back to the window they were cut from, both strands (reverse-strand hits
            lie above 2^32), Default (maxSpan) and Illumina (SMEM: 35-bit packed list entries) presets;
      (iv)  the ambiguity boundary (segment.h:316-349, parameter.h:686-689): the 40-mer with 100 occurrences yields exactly
            its 100 planted positions as seeds, the one with 101 is skipped -- on both strands."""
    import torch
    import ma_amd
    L = ma_amd.lib()
    lens = np.array(GRCH38, dtype=np.uint64)
    F = int(lens.sum())
    g = torch.empty(F, dtype=torch.uint8, device="cuda")
    assert L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(F), C.c_int32(1), C.c_void_p(g.data_ptr())) == 0
    # ---- (iv) two probes, planted at fixed positions spread over the genome (well inside contigs, 1 kb apart at least)
    rng = np.random.default_rng(404)
    probes = [rng.integers(0, 4, size=40, dtype=np.uint8) for _ in range(2)]
    starts = np.cumsum(np.concatenate([[0], lens[:-1]])).astype(np.int64)
    slots = []
    for c in range(len(lens)):
        span = int(lens[c]) - 200000
        k = 9 if c < 9 else 8  # 9 * 9 + 15 * 8 = 201 slots
        slots += [int(starts[c]) + 100000 + (span // k) * j for j in range(k)]
    assert len(slots) == 201
    pos100, pos101 = sorted(slots[:100]), sorted(slots[100:])
    for pos, pr in ((pos100, probes[0]), (pos101, probes[1])):
        pt = torch.from_numpy(pr).cuda()
        for p in pos:
            g[p:p + 40] = pt
    idx = ma_amd.Index.build_device(lens, g.data_ptr())
    d = idx.download()
    n = int(d["ref_len"])
    assert n == 2 * F and n > 2 ** 32
    primary = int(d["primary"])
    # ---- (i) occ counters: recount on the device, block by block
    bwt = torch.from_numpy(d["bwt"].view(np.int32)).cuda()
    nblk = bwt.numel() // 16
    blocks = bwt[: nblk * 16].view(nblk, 16)
    lut = torch.zeros(4, 256, dtype=torch.int16)
    for v in range(256):
        for k in range(4):
            lut[(v >> (2 * k)) & 3, v] += 1
    lut = lut.cuda()
    per = torch.zeros(nblk, 4, dtype=torch.int64, device="cuda")
    step = 1 << 22
    for lo in range(0, nblk, step):
        by = blocks[lo:lo + step, 8:].contiguous().view(torch.uint8).view(-1, 32).long()
        for c in range(4):
            per[lo:lo + step, c] = lut[c][by].sum(dim=1)
    # the text has n symbols ('$' is not stored): the padding of the last block is zero bits (= A) and must not be counted
    stored = blocks[:, :8].contiguous().view(torch.int64).view(nblk, 4)
    want = torch.cumsum(per, dim=0) - per  # exclusive
    full = n // 128  # blocks entirely inside the text
    assert torch.equal(stored[: full + 1], want[: full + 1]), "occ counters differ from a recount of the packed BWT"
    tail = n - full * 128
    tot = want[full].clone()
    if tail:
        w = d["bwt"][full * 16 + 8:full * 16 + 16]
        for j in range(tail):
            tot[int((int(w[j >> 4]) >> ((~j & 15) << 1)) & 3)] += 1
    L2 = d["L2"].astype(np.int64)
    assert [int(L2[c + 1] - L2[c]) for c in range(4)] == [int(x) for x in tot.tolist()], "L2 is not the symbol histogram"
    # symbol counts of the BWT = symbol counts of the text T . revcomp(T): A <-> T, C <-> G
    hist = torch.bincount(g.view(-1).to(torch.int64), minlength=4).tolist() if F < 2 ** 31 else [int((g == c).sum()) for c in range(4)]
    assert [hist[c] + hist[3 - c] for c in range(4)] == [int(x) for x in tot.tolist()]
    del bwt, blocks, per, stored, want
    torch.cuda.empty_cache()
    # ---- (ii) sampled suffix-array rows
    gh = g.cpu().numpy()
    sa = d["sa"]

    def text(p):  # T[p] for arrays of positions (0 <= p < n)
        p = np.asarray(p, dtype=np.int64)
        fw = p < F
        out = np.empty(p.shape, dtype=np.uint8)
        out[fw] = gh[p[fw]]
        out[~fw] = 3 - gh[n - 1 - p[~fw]]
        return out

    ks = np.sort(rng.integers(1, len(sa) - 1, size=100000))
    a, b = sa[ks].astype(np.int64), sa[ks + 1].astype(np.int64)
    assert a.min() >= 0 and max(a.max(), b.max()) < n and (a > 2 ** 32).any()
    W = 256
    undecided = np.ones(len(ks), dtype=bool)
    off = 0
    while undecided.any() and off < 65536:
        ii = np.nonzero(undecided)[0]
        pa = a[ii, None] + off + np.arange(W)[None, :]
        pb = b[ii, None] + off + np.arange(W)[None, :]
        ea, eb = pa >= n, pb >= n  # past the end of the text: '$', smaller than every base
        ta = np.where(ea, -1, text(np.minimum(pa, n - 1)).astype(np.int16))
        tb = np.where(eb, -1, text(np.minimum(pb, n - 1)).astype(np.int16))
        ne = ta != tb
        first = ne.argmax(axis=1)
        has = ne.any(axis=1)
        rows = np.arange(len(ii))
        assert np.all(ta[rows[has], first[has]] < tb[rows[has], first[has]]), "sampled suffixes out of order"
        undecided[ii[has]] = False
        off += W
    assert not undecided.any()
    # BWT symbol of row 32 k = T[sa[k] - 1] (rows with sa == 0 hold '$': the primary row)
    rows_ = ks.astype(np.int64) * 32
    keep = rows_ != primary
    xx = rows_[keep] - (rows_[keep] > primary)
    w = d["bwt"][(xx >> 7) * 16 + 8 + ((xx & 127) >> 4)]
    sym = (w >> (((~(xx & 127)) & 15) << 1)) & 3
    assert np.array_equal(sym.astype(np.uint8), text(a[keep] - 1)), "BWT symbols of sampled rows"
    del d
    # ---- (iii) error-free reads map back, both presets, 150 bp and 10 kb
    for preset, n_reads, rl in (("default", 200000, 150), ("illumina", 100000, 150), ("default", 2000, 10000), ("illumina", 300, 2000)):
        codes, offs, nb = _synth_reads(L, idx, 11, n_reads, rl)
        bt = ma_amd.Batch(idx, ma_amd.Params.preset(preset), n_reads, nb + 64)
        bt.set_reads_device(codes.data_ptr(), offs.data_ptr(), n_reads, nb)
        bt.align()
        bt.sync()
        moff, alns, ops = bt.mapq_alignments()
        assert bt.counts()["aligned_reads"] == n_reads, (preset, rl)
        first = alns[moff[:-1].astype(np.int64)]
        assert np.all(first["score"] == 2 * rl) and np.all(first["begin_q"] == 0) and np.all(first["end_q"] == rl), (preset, rl)
        assert np.all(first["end_ref"] - first["begin_ref"] == rl)
        assert (first["begin_ref"] > 2 ** 32).sum() > 0.2 * n_reads  # reverse-strand reads: coordinates above 2^32
        rc = codes[: n_reads * rl].cpu().numpy().reshape(n_reads, rl)
        for i in range(0, n_reads, max(1, n_reads // 400)):
            br = int(first["begin_ref"][i])
            assert np.array_equal(text(np.arange(br, br + rl)), rc[i]), (preset, rl, i)
        bt.close()
        del codes, offs
    # ---- (iv) the ambiguity boundary, forward and reverse strand
    P = ma_amd.Params.preset("default")
    reads = [probes[0], probes[1], (3 - probes[0])[::-1].copy(), (3 - probes[1])[::-1].copy()]
    bt = ma_amd.Batch(idx, P, 4, 4 * 40 + 64)
    bt.set_reads(reads)
    bt.seed()
    bt.extract()
    bt.sync()
    soff, segs = bt.segments()
    doff, seeds = bt.seeds()
    for r in range(4):
        sg = segs[int(soff[r]):int(soff[r + 1])]
        assert len(sg) == 1 and int(sg[0]["q_size"]) == 39, (r, sg)
        assert int(sg[0]["sa_size"]) == (100 if r % 2 == 0 else 101), (r, sg)
    for r, pos in ((0, pos100), (2, pos100)):
        sd = seeds[int(doff[r]):int(doff[r + 1])]
        assert len(sd) == 100 and np.all(sd["len"] == 40) and np.all(sd["ambiguity"] == 100)
        # seeds are reported on the forward strand with an orientation flag (segment.h:99-105)
        got = np.sort(np.where(sd["on_forward"] != 0, sd["r_start"], sd["r_start"] - 39).astype(np.int64))
        assert np.array_equal(got, np.array(pos, dtype=np.int64)), (r, got[:5], pos[:5])
    for r in (1, 3):
        assert int(doff[r + 1]) == int(doff[r]), "a segment with 101 occurrences must be skipped"
    bt.close()
    idx.close()


def test_c1_ecoli_like_vs_compiled_reference(gpu_device, tmp_path):
    """configs[0] (C1) at its stated size: 1 k x 150 bp reads (0.5 % substitutions, seed 11) vs `ecoli_like` (one contig of
    4 641 652 nt, seed 1).  The doubled text (9.28 Mnt) lies just BELOW the 10 Mnt switch, so the drop-all and SoC-score
    heuristics are off (binarySeeding.cpp:172-175, stripOfConsideration.cpp:21-23).  The compiled reference builds ITS OWN index
    of the contig; every stage record of every read must equal the GPU path's."""
    import torch
    import ma_amd
    from test_gpu_parity import compare_reads, gpu_pipeline
    ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(ref_dump):
        pytest.skip("oracle/_ref not present on this box")
    L = ma_amd.lib()
    F = 4641652
    g = torch.empty(F, dtype=torch.uint8, device="cuda")
    assert L.ma_synth_genome_device(C.c_uint64(1), C.c_uint64(F), C.c_int32(0), C.c_void_p(g.data_ptr())) == 0
    idx = ma_amd.Index.build_device(np.array([F], dtype=np.uint64), g.data_ptr())
    assert idx.sizes()[2] == 2 * F < 10000000
    codes, offs, nb = _synth_reads(L, idx, 11, 1000, 150, sub=0.005)
    oh = offs.cpu().numpy()
    ch = codes.cpu().numpy()
    reads = [ch[int(oh[i]):int(oh[i + 1])].copy() for i in range(1000)]
    reads += [reads[0][:12].copy(), np.full(150, 4, dtype=np.uint8)]  # shorter than a seed, all N
    case = str(tmp_path / "c1.case")
    write_case(case, [g.cpu().numpy()], reads)
    for preset in ("default", "illumina"):
        subprocess.check_call([ref_dump, "pipe", case, preset, "1", str(tmp_path / "ref.pipe")], stdout=subprocess.DEVNULL)
        want = parse_pipe_dump(str(tmp_path / "ref.pipe"))
        got, counters, counts = gpu_pipeline(idx, preset, 1, reads)
        compare_reads(got, want)
        assert counts["aligned_reads"] == sum(1 for w in want if w["mq"]) >= 990
    idx.close()


def test_host_to_host_io_through_page_locked_arrays(gpu_device):
    """The I/O path of bench.py's headline leg: reads handed over as ONE page-locked host array + CSR offsets
    (Batch.set_reads_flat over ma_host_alloc memory), the MappingQuality records downloaded into caller-owned page-locked arrays
    (mapq_alignments_into) -- identical to the list-of-reads / numpy path, too small result arrays are reported (None), and
    the same batch object serves a second, different read set."""
    import ma_amd
    from ma_testlib import rand_genome, sample_reads
    g = rand_genome(77, [300000, 200000], repeat_unit=200, repeat_copies=40, repeat_div=0.05)
    idx = ma_amd.Index.build(g)
    P = ma_amd.Params.preset("default")
    sets = [sample_reads(g, 3000, 150, 5, sub=0.01) + sample_reads(g, 20, 3000, 6, sub=0.02, ins=0.01, dele=0.01),
            sample_reads(g, 1000, 100, 7, sub=0.03, n_rate=0.01) + [np.zeros(0, dtype=np.uint8)]]
    nb = max(sum(len(r) for r in s) for s in sets)
    nr = max(len(s) for s in sets)
    b1 = ma_amd.Batch(idx, P, nr, nb + 64)
    b2 = ma_amd.Batch(idx, P, nr, nb + 64)
    codes = ma_amd.HostArray(nb + 64, np.uint8)
    offs = ma_amd.HostArray(nr + 1, np.uint64)
    out_off = ma_amd.HostArray(nr + 1, np.uint64)
    small = (ma_amd.HostArray(nr + 1, np.uint64), ma_amd.HostArray(8, ma_amd.ALIGNMENT_DT), ma_amd.HostArray(8, np.uint64))
    for reads in sets:
        b1.set_reads(reads)
        b1.align()
        b1.sync()
        woff, walns, wops = b1.mapq_alignments()
        n = len(reads)
        o = np.zeros(n + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(r) for r in reads])
        codes.a[: int(o[n])] = np.concatenate([np.asarray(r, dtype=np.uint8) for r in reads]) if int(o[n]) else []
        offs.a[: n + 1] = o
        b2.set_reads_flat(codes.ptr, offs.ptr, n)
        b2.align()
        b2.sync()
        assert b2.mapq_alignments_into(*small) is None  # too small: nothing is written, the caller grows its arrays
        c = b2.counts()
        out_alns = ma_amd.HostArray(c["alignments"] + 1, ma_amd.ALIGNMENT_DT)
        out_ops = ma_amd.HostArray(2 * c["ops_cap"] + 2, np.uint64)
        assert b2.mapq_alignments_into(out_off, out_alns, out_ops) == c
        na = int(woff[n])
        assert np.array_equal(out_off.a[: n + 1], woff)
        assert out_alns.a[:na].tobytes() == walns[:na].tobytes()
        nops = int(walns["ops_off"][na - 1] + walns["n_ops"][na - 1]) if na else 0
        assert np.array_equal(out_ops.a[: 2 * nops], wops[: 2 * nops])
        out_alns.close()
        out_ops.close()
    for h in (codes, offs, out_off) + small:
        h.close()
    b1.close()
    b2.close()
    idx.close()
