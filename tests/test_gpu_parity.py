"""Parity tests proper: the HIP path (through the C ABI) against the oracle and against the golden
vectors produced by the real reference.  Bit-exact on every integer output."""
import os

import numpy as np
import pytest

from ma_testlib import (ROOT, gunzip_to, read_case, read_ksw_cases, parse_pipe_dump, parse_ksw_dump, OrIndex, or_params,
                        rand_genome, sample_reads, rand_ksw_cases, or_ksw, revcomp, write_case, KSW_EXTZ, KSW_RIGHT, KSW_REV)

pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, "tests", "golden")


def gpu_params(preset, seed):
    import ma_amd
    P = ma_amd.Params.preset("illumina" if preset.startswith("illumina") else "default")
    if preset.endswith("+mems"):  # "Seeding Technique" = MEMs (binarySeeding.h:460-537, selected by no preset)
        P.seeding_technique = 2
    P.srand_seed = seed
    return P


def gpu_pipeline(index, preset, seed, reads, stages=True):
    import ma_amd
    P = gpu_params(preset, seed)
    b = ma_amd.Batch(index, P, max(len(reads), 1), sum(len(r) for r in reads) + 64)
    b.set_reads(reads)
    b.seed()
    b.extract()
    b.chain()
    b.dp()
    b.sync()
    out = []
    soff, segs = b.segments()
    doff, seeds = b.seeds()
    hoff, hsoff, hsoc, hseeds = b.hsets()
    aoff, alns, ops = b.alignments()
    moff, malns, mops = b.mapq_alignments()

    def seedt(s):
        return (int(s["q_start"]), int(s["len"]), int(s["r_start"]), int(s["ambiguity"]), int(s["on_forward"]),
                int(s["delta"]))

    for r in range(len(reads)):
        d = dict(len=len(reads[r]))
        d["segs"] = [tuple(int(x) for x in s) for s in segs[int(soff[r]):int(soff[r + 1])]]
        d["seeds"] = [seedt(s) for s in seeds[int(doff[r]):int(doff[r + 1])]]
        d["hsets"] = []
        for h in range(int(hoff[r]), int(hoff[r + 1])):
            d["hsets"].append(dict(soc=int(hsoc[h]), seeds=[seedt(s) for s in hseeds[int(hsoff[h]):int(hsoff[h + 1])]]))
        d["alns"] = []
        for a in alns[int(aoff[r]):int(aoff[r + 1])]:
            o = int(a["ops_off"])
            d["alns"].append(dict(bref=int(a["begin_ref"]), eref=int(a["end_ref"]), bq=int(a["begin_q"]),
                                  eq=int(a["end_q"]), score=int(a["score"]), soc=int(a["soc_index"]),
                                  ops=[(int(ops[2 * (o + k)]), int(ops[2 * (o + k) + 1])) for k in range(int(a["n_ops"]))]))
        d["mq"] = []
        for a in malns[int(moff[r]):int(moff[r + 1])]:
            d["mq"].append(dict(bref=int(a["begin_ref"]), eref=int(a["end_ref"]), bq=int(a["begin_q"]), eq=int(a["end_q"]),
                                score=int(a["score"]), secondary=int(a["secondary"]),
                                supplementary=int(a["supplementary"]), mapq=float(a["mapq"])))
        out.append(d)
    counters = b.counters()
    counts = b.counts()
    b.close()
    return out, counters, counts


def compare_reads(got, want, what=("segs", "seeds", "hsets", "alns", "mq")):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        for k in what:
            if k == "mq":
                assert len(g[k]) == len(w[k]), "read %d: mq count %d vs %d" % (i, len(g[k]), len(w[k]))
                for a, b in zip(g[k], w[k]):
                    for f in ("bref", "eref", "bq", "eq", "score", "secondary", "supplementary"):
                        assert a[f] == b[f], "read %d mq field %s: %s vs %s" % (i, f, a, b)
                    assert float("%.17g" % a["mapq"]) == b["mapq"], "read %d mapq %r vs %r" % (i, a["mapq"], b["mapq"])
            else:
                assert g[k] == w[k], "read %d stage %s differs:\n got  %s\n want %s" % (i, k, g[k][:6], w[k][:6])


@pytest.fixture(scope="module")
def small(tmp_path_factory, gpu_device):
    import ma_amd
    d = tmp_path_factory.mktemp("gpu")
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(d / "small.case"))
    contigs, reads, _ = read_case(case)
    oidx = OrIndex.build(contigs)
    gidx = ma_amd.Index.from_arrays(**oidx.arrays())
    return dict(contigs=contigs, reads=reads, oidx=oidx, gidx=gidx, dir=d)


def test_extend_backward_and_bwt_sa(small):
    oidx, gidx = small["oidx"], small["gidx"]
    A = oidx.arrays()
    n, L2 = A["ref_len"], A["L2"]
    rng = np.random.default_rng(5)
    iks, cs = [], []
    # intervals reached by backward search of random reads (covers primary / block boundaries) + raw ones
    for r in small["reads"][:60]:
        q = r[r < 4]
        if len(q) < 2:
            continue
        c0 = int(q[-1])
        ik = np.array([L2[c0] + 1, L2[3 - c0] + 1, L2[c0 + 1] - L2[c0]], dtype=np.int64)
        for j in range(len(q) - 2, max(len(q) - 14, -1), -1):
            for c in range(5):
                iks.append(ik.copy())
                cs.append(c)
            ik = oidx.extend_backward(ik.reshape(1, 3), [int(q[j])])[0]
            if ik[2] <= 0:
                break
    for _ in range(500):
        s = int(rng.integers(1, n))
        sz = int(rng.integers(1, min(2000, n + 1 - s) + 1))
        iks.append(np.array([s, int(rng.integers(1, n)), sz], dtype=np.int64))
        cs.append(int(rng.integers(0, 5)))
    iks.append(np.array([1, 1, n], dtype=np.int64))  # whole range incl. primary
    cs.append(2)
    iks = np.array(iks, dtype=np.int64)
    cs = np.array(cs, dtype=np.uint8)
    want = oidx.extend_backward(iks, cs)
    got = gidx.extend_backward(iks, cs)
    assert np.array_equal(got, want)
    rows = np.concatenate([rng.integers(1, n + 1, size=3000), [1, n, A["primary"], 32, 31, 33]]).astype(np.int64)
    assert np.array_equal(gidx.bwt_sa(rows), oidx.bwt_sa(rows))


def test_pack_extract(small):
    """ma_pack_extract = Pack::vExtract (pack.h:1147-1236): forward ranges, reverse-strand ranges (complement of the
    mirrored forward base), empty ranges; bridging and out-of-range requests fail like the reference throws."""
    import ma_amd
    gidx = small["gidx"]
    fwd = np.concatenate(small["contigs"]).astype(np.uint8)
    F = len(fwd)
    text = np.concatenate([fwd, (3 - fwd[::-1]).astype(np.uint8)])
    rng = np.random.default_rng(17)
    b = np.concatenate([rng.integers(0, F - 700, 40), rng.integers(F, 2 * F - 700, 40), [0, F, 2 * F - 1, 5, F - 1]])
    e = b + np.concatenate([rng.integers(1, 700, 80), [1, 1, 1, 0, 1]])
    got = gidx.extract(b, e)
    for i in range(len(b)):
        assert np.array_equal(got[i], text[int(b[i]):int(e[i])]), "range %d" % i
    for bb, ee in ((F - 3, F + 3), (10, 5), (2 * F - 1, 2 * F + 1)):
        with pytest.raises(ma_amd.MaError, match="vExtractSubsection"):
            gidx.extract([bb], [ee])


def test_pipeline_long_read_presets_vs_oracle(small):
    """PacBio / Nanopore presets (parameter.h:1096-1104): up to 100 supplementary alignments, at least 5 SoCs, SMEMs."""
    import ma_amd
    # f4.case has the genome of small.case; its last reads carry small inversions -> supplementary alignments
    _, f4reads, _ = read_case(gunzip_to(os.path.join(G, "f4.case.gz"), str(small["dir"] / "f4.case")))
    reads = small["reads"][:60] + f4reads[-16:]
    for technique in (0, 1):
        P = ma_amd.Params.preset("default")
        op = or_params("default", 1)
        for prm in (P, op):
            prm.max_supplementary, prm.min_num_soc, prm.seeding_technique, prm.srand_seed = 100, 5, technique, 1
        b = ma_amd.Batch(small["gidx"], P, len(reads), sum(len(r) for r in reads) + 64)
        b.set_reads(reads)
        b.align()
        b.sync()
        moff, malns, mops = b.mapq_alignments()
        res = small["oidx"].align(reads, op)
        assert np.array_equal(moff, res["mq_off"])
        for f in ("begin_ref", "end_ref", "begin_q", "end_q", "score", "secondary", "supplementary"):
            assert np.array_equal(malns[f], res["mq"][f]), f
        assert np.array_equal(malns["mapq"].view(np.uint64), res["mq"]["mapq"].view(np.uint64))
        assert int(malns["supplementary"].sum()) > 0


@pytest.mark.parametrize("scoring", [(3, 5, 6, 3, 30, 2), (1, 3, 5, 2, 24, 1), (2, 4, 24, 1, 4, 2)])
def test_pipeline_other_scoring_schemes_vs_oracle(small, scoring):
    """The whole path under other global scoring parameters (they enter the SoC thresholds, the gap-cost estimation of
    Harmonization, every DP call, the alignment scores and MappingQuality): NeedlemanWunsch and MappingQuality
    records against the oracle.  (The oracle's kswcpp is pinned against the reference for these schemes; the reference's
    pipeline is not a well-defined function of them, see tests/test_oracle_vs_ref.py: one pGlobalParams per unit.)"""
    import ma_amd
    _, f4reads, _ = read_case(gunzip_to(os.path.join(G, "f4.case.gz"), str(small["dir"] / "f4.case")))
    reads = small["reads"] + f4reads[-8:] + f4reads[:40]
    P = ma_amd.Params.preset("default")
    op = or_params("default", 1)
    for prm in (P, op):
        prm.match, prm.mismatch, prm.gap, prm.extend, prm.gap2, prm.extend2 = scoring
        prm.srand_seed = 1
    b = ma_amd.Batch(small["gidx"], P, len(reads), sum(len(r) for r in reads) + 64)
    b.set_reads(reads)
    b.align()
    b.sync()
    res = small["oidx"].align(reads, op)
    aoff, alns, ops = b.alignments()
    assert np.array_equal(aoff, res["aln_off"])
    for f in ("begin_ref", "end_ref", "begin_q", "end_q", "score", "soc_index", "n_ops"):
        assert np.array_equal(alns[f], res["alns"][f]), f
    for g, o in zip(alns, res["alns"]):
        assert np.array_equal(ops[2 * int(g["ops_off"]):2 * int(g["ops_off"] + g["n_ops"])],
                              res["ops"][2 * int(o["ops_off"]):2 * int(o["ops_off"] + o["n_ops"])])
    moff, malns, _ = b.mapq_alignments()
    assert np.array_equal(moff, res["mq_off"])
    for f in ("begin_ref", "end_ref", "score", "secondary", "supplementary"):
        assert np.array_equal(malns[f], res["mq"][f]), f
    assert np.array_equal(malns["mapq"].view(np.uint64), res["mq"]["mapq"].view(np.uint64))


# reference outputs for ksw.case under the presets' scoring and under make_golden.py's KSW_SCORINGS
KSW_GOLDEN = [("ksw_ref.out.gz", None), ("ksw_ref.sc0.out.gz", (3, 5, 6, 3, 30, 2)), ("ksw_ref.sc1.out.gz", (1, 3, 5, 2, 24, 1)),
              ("ksw_ref.sc2.out.gz", (2, 4, 24, 1, 4, 2)), ("ksw_ref.sc3.out.gz", (5, 4, 2, 1, 40, 1))]


@pytest.mark.parametrize("name,scoring", KSW_GOLDEN)
def test_ksw_golden_cases(gpu_device, tmp_path, name, scoring):
    import ma_amd
    case = gunzip_to(os.path.join(G, "ksw.case.gz"), str(tmp_path / "ksw.case"))
    ref = parse_ksw_dump(os.path.join(G, name))
    cases = read_ksw_cases(case)
    P = ma_amd.Params.preset("default")
    if scoring:
        P.match, P.mismatch, P.gap, P.extend, P.gap2, P.extend2 = scoring
    ez, cigs = ma_amd.ksw_batch(P, cases)
    for i, w in enumerate(ref):
        for f in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "reach_end", "n_cigar"):
            assert int(ez[f][i]) == w[f], "case %d field %s: %d vs %d (qlen %d tlen %d w %d zdrop %d flag %d)" % (
                i, f, int(ez[f][i]), w[f], len(cases[i][0]), len(cases[i][1]), cases[i][2], cases[i][3], cases[i][4])
        assert list(cigs[i]) == w["cigar"], "case %d cigar" % i


def test_ksw_random_vs_oracle(gpu_device):
    import ma_amd
    cases = rand_ksw_cases(1500, 4242, max_len=260) + rand_ksw_cases(20, 4243, long_frac=1.0)
    # degenerate shapes
    cases += [(np.array([1], dtype=np.uint8), np.array([1, 2, 3] * 30, dtype=np.uint8), 512, 200, 0x40),
              (np.array([0, 1, 2, 3] * 10, dtype=np.uint8), np.array([2], dtype=np.uint8), 20, -1, 0),
              (np.array([4] * 20, dtype=np.uint8), np.array([4] * 17, dtype=np.uint8), 20, -1, 0)]
    P = ma_amd.Params.preset("default")
    ez, cigs = ma_amd.ksw_batch(P, cases)
    op = or_params()
    for i, (q, t, w, zd, fl) in enumerate(cases):
        oez, ocig = or_ksw(op, q, t, w, zd, fl)
        for f in oez.dtype.names:
            assert int(ez[f][i]) == int(oez[f]), "case %d field %s (qlen %d tlen %d w %d zdrop %d flag %d)" % (
                i, f, len(q), len(t), w, zd, fl)
        assert np.array_equal(cigs[i], ocig), "case %d cigar" % i


def ext_shaped_cases(n, seed, ql_lo=1, ql_hi=250):
    """Jobs shaped like NeedlemanWunsch's extensions: short query, long padded target, w = 512, zdrop = 200; plus
    the small global gap fills (w >= qlen + tlen) that the extension kernel also takes."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        kind = rng.random()
        ql = int(rng.integers(ql_lo, ql_hi)) if kind < 0.85 else int(rng.integers(1, 7))
        if kind < 0.85:
            tl = int(rng.integers(max(1, ql - 20), ql + 1100))
            t = rng.integers(0, 4, size=tl, dtype=np.uint8)
            if rng.random() < 0.3:  # low complexity target: late maxima are plausible
                unit = rng.integers(0, 4, size=int(rng.integers(1, 6)), dtype=np.uint8)
                t = np.tile(unit, tl // len(unit) + 1)[:tl].copy()
            q = t[:min(ql, tl)].copy()
            if len(q) < ql:
                q = np.concatenate([q, rng.integers(0, 4, size=ql - len(q), dtype=np.uint8)])
            er = rng.choice([0.0, 0.02, 0.08, 0.25, 0.75])
            mut = rng.random(ql) < er
            q[mut] = (q[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
            if rng.random() < 0.3 and ql > 8:  # an indel
                k = int(rng.integers(1, ql - 1))
                q = np.concatenate([q[:k], q[k + int(rng.integers(1, 4)):], rng.integers(0, 4, size=3, dtype=np.uint8)])[:ql]
            if rng.random() < 0.1:
                q[int(rng.integers(0, len(q)))] = 4
            flag = KSW_EXTZ if rng.random() < 0.5 else (KSW_EXTZ | KSW_RIGHT | KSW_REV)
            cases.append((q, t, int(rng.choice([512, 512, 300, 260])), int(rng.choice([200, 200, 40, -1])), flag))
        else:
            tl = int(rng.integers(1, 120))
            t = rng.integers(0, 4, size=tl, dtype=np.uint8)
            q = rng.integers(0, 4, size=ql, dtype=np.uint8)
            cases.append((q, t, max(20, abs(tl - ql) + 10), -1, 0))
    return cases


def test_ksw_pipeline_semantics_vs_oracle(gpu_device):
    """ma_ksw_ext_batch (packed extension kernel, early stop) against kswcpp as restated by the oracle, on what the
    pipeline reads: max, max_q, max_t and the cigar."""
    import ma_amd
    # + the query lengths around the capacity limits of the one- and two-slot rings (113 / 126 / 241 / 254 cells)
    cases = (ext_shaped_cases(3000, 99) + rand_ksw_cases(600, 777, max_len=200) + ext_shaped_cases(900, 98, 108, 132)
             + ext_shaped_cases(500, 97, 236, 260))
    P = ma_amd.Params.preset("default")
    ez, cigs = ma_amd.ksw_batch(P, cases, pipeline_semantics=True)
    op = or_params()
    n_ext = 0
    for i, (q, t, w, zd, fl) in enumerate(cases):
        oez, ocig = or_ksw(op, q, t, w, zd, fl)
        what = ("max", "max_q", "max_t") if fl & KSW_EXTZ else ()
        for f in what:
            assert int(ez[f][i]) == int(oez[f]), "case %d field %s: %d vs %d (qlen %d tlen %d w %d zdrop %d flag %d)" % (
                i, f, int(ez[f][i]), int(oez[f]), len(q), len(t), w, zd, fl)
        assert np.array_equal(cigs[i], ocig), "case %d cigar (qlen %d tlen %d w %d zdrop %d flag %d)" % (
            i, len(q), len(t), w, zd, fl)
        n_ext += 1 if fl & KSW_EXTZ else 0
    assert n_ext > 2000


@pytest.mark.parametrize("scoring", [(3, 5, 6, 3, 30, 2), (1, 3, 5, 2, 24, 1), (2, 4, 24, 1, 4, 2), (5, 4, 2, 1, 40, 1)])
def test_ksw_other_scoring_schemes(gpu_device, scoring):
    """match, mismatch, gap, extend, gap2, extend2 other than the presets' 2/4/4/2/24/1 (incl. the swapped order of the
    two gap models, kswcpp_core.h:366-376): every ez field of the exact kernel and the pipeline semantics."""
    import ma_amd
    P = ma_amd.Params.preset("default")
    op = or_params()
    for prm in (P, op):
        prm.match, prm.mismatch, prm.gap, prm.extend, prm.gap2, prm.extend2 = scoring
    cases = rand_ksw_cases(500, 900 + scoring[0], max_len=220) + rand_ksw_cases(6, 950, long_frac=1.0) + ext_shaped_cases(700, 55)
    ez, cigs = ma_amd.ksw_batch(P, cases)
    ez2, cigs2 = ma_amd.ksw_batch(P, cases, pipeline_semantics=True)
    for i, (q, t, w, zd, fl) in enumerate(cases):
        oez, ocig = or_ksw(op, q, t, w, zd, fl)
        for f in oez.dtype.names:
            assert int(ez[f][i]) == int(oez[f]), "case %d field %s (qlen %d tlen %d w %d zdrop %d flag %d)" % (
                i, f, len(q), len(t), w, zd, fl)
        assert np.array_equal(cigs[i], ocig), "case %d cigar" % i
        if fl & KSW_EXTZ:
            for f in ("max", "max_q", "max_t"):
                assert int(ez2[f][i]) == int(oez[f]), "pipeline semantics: case %d field %s" % (i, f)
        assert np.array_equal(cigs2[i], ocig), "pipeline semantics: case %d cigar" % i


@pytest.mark.parametrize("preset", ["default", "illumina", "default+mems"])
def test_pipeline_vs_compiled_reference_direct(gpu_device, tmp_path, preset):
    """No oracle in the loop: when the compiled reference travelled with the repository (oracle/_ref, built from
    /root/reference by oracle/Makefile.ref), its own modules and the GPU path align the same reads of a genome large enough
    for the heuristics (> 10 Mnt doubled), with repeats, N bases, long reads: every stage record must be identical."""
    import subprocess
    import ma_amd
    ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(ref_dump):
        pytest.skip("oracle/_ref not present on this box")
    g = rand_genome(19, [2700000, 1400000, 1000000], repeat_unit=300, repeat_copies=250, repeat_div=0.08)
    reads = (sample_reads(g, 2500, 150, 131, sub=0.01) + sample_reads(g, 300, 150, 132, sub=0.06, n_rate=0.01)
             + sample_reads(g, 20, 5000, 133, sub=0.005, ins=0.003, dele=0.003) + sample_reads(g, 40, 150, 135, random_frac=1.0)
             + sample_reads(g, 200, 100, 136, sub=0.03) + sample_reads(g, 2, 25000, 134, sub=0.03, ins=0.03, dele=0.04))
    case = str(tmp_path / "c.case")
    write_case(case, g, reads)
    subprocess.check_call([ref_dump, "pipe", case, preset, "3", str(tmp_path / "ref.pipe")], stdout=subprocess.DEVNULL)
    want = parse_pipe_dump(str(tmp_path / "ref.pipe"))
    got, counters, counts = gpu_pipeline(ma_amd.Index.build(g), preset, 3, reads)
    compare_reads(got, want)
    assert counts["aligned_reads"] == sum(1 for w in want if w["mq"])


@pytest.mark.parametrize("preset,seed,name", [("default", 1, "small_ref.default.pipe"),
                                              ("illumina", 1, "small_ref.illumina.pipe"),
                                              ("default", 7, "small_ref.default.seed7.pipe"),
                                              ("default+mems", 1, "small_ref.mems.pipe")])
def test_pipeline_vs_reference_golden(small, preset, seed, name):
    want = parse_pipe_dump(os.path.join(G, name + ".gz"))
    got, counters, counts = gpu_pipeline(small["gidx"], preset, seed, small["reads"])
    compare_reads(got, want)
    assert counts["aligned_reads"] == sum(1 for w in want if w["mq"])


def test_pipeline_counters_match_oracle(small):
    res = small["oidx"].align(small["reads"], or_params("default", 1))
    got, counters, counts = gpu_pipeline(small["gidx"], "default", 1, small["reads"])
    c = res["counters"]
    # extend_backward steps / distinct occ blocks: runs that start at a centre take their first steps from the K-mer table
    assert 0 < int(counters[0]) <= int(c[0])
    assert 0 < int(counters[1]) <= int(c[1])
    # LF steps: the device walks to the next row of its denser SA sample (every 8th row), the oracle to the reference's every 32nd
    assert 0 < int(counters[2]) <= int(c[2])
    assert int(counters[3]) == int(c[3])  # SA rows
    # DP band cells: the pipeline stops an extension once no later diagonal can raise ez.max (ksw_reg.h)
    assert 0 < int(counters[4]) <= int(c[4])
    assert int(counters[5]) == int(c[5])  # ksw calls


def test_seed_staging_overflow_is_retried_with_full_capacity(small, monkeypatch):
    """Long reads stage fewer segments per lane than the worst case; a read that overflows makes the stage run again
    with the worst-case capacity (MA_SEED_STAGE_CAP forces the first attempt to be too small)."""
    want = parse_pipe_dump(os.path.join(G, "small_ref.default.pipe.gz"))
    monkeypatch.setenv("MA_SEED_STAGE_CAP", "2")
    got, counters, counts = gpu_pipeline(small["gidx"], "default", 1, small["reads"])
    compare_reads(got, want)


def test_empty_and_ragged_batches(small):
    import ma_amd
    got, _, counts = gpu_pipeline(small["gidx"], "default", 1, [])
    assert got == [] and counts["alignments"] == 0
    reads = [np.zeros(0, dtype=np.uint8), np.array([1], dtype=np.uint8), np.full(40, 4, dtype=np.uint8),
             small["reads"][0], small["reads"][0][:20]]
    oreads = [r for r in reads]
    res = small["oidx"].align([r if len(r) else np.zeros(0, dtype=np.uint8) for r in oreads], or_params())
    got, _, _ = gpu_pipeline(small["gidx"], "default", 1, reads)
    for r in range(len(reads)):
        want_segs = [tuple(int(x) for x in s) for s in res["segs"][int(res["seg_off"][r]):int(res["seg_off"][r + 1])]]
        assert got[r]["segs"] == want_segs
        assert len(got[r]["alns"]) == int(res["aln_off"][r + 1] - res["aln_off"][r])


def test_index_build_on_gpu_matches_reference_files(small, tmp_path):
    import ma_amd
    gidx = ma_amd.Index.build(small["contigs"])
    d = gidx.download()
    ref_bwt = np.fromfile(gunzip_to(os.path.join(G, "small_ref.bwt.gz"), str(tmp_path / "r.bwt")), dtype=np.uint8)
    ref_sa = np.fromfile(gunzip_to(os.path.join(G, "small_ref.sa.gz"), str(tmp_path / "r.sa")), dtype=np.uint8)
    ref_pac = np.fromfile(gunzip_to(os.path.join(G, "small_ref.pac.gz"), str(tmp_path / "r.pac")), dtype=np.uint8)
    # .bwt = primary(8) L2[1..4](32) words ; .sa = primary(8) L2(32) intv(4) seq_len(8) sa[1..]
    hdr = np.concatenate([np.array([d["primary"]], dtype=np.int64).view(np.uint8), d["L2"][1:5].view(np.uint8)])
    assert np.array_equal(np.concatenate([hdr, d["bwt"].view(np.uint8)]), ref_bwt)
    sa_hdr = np.concatenate([hdr, np.array([32], dtype=np.int32).view(np.uint8),
                             np.array([d["ref_len"]], dtype=np.uint64).view(np.uint8)])
    assert np.array_equal(np.concatenate([sa_hdr, d["sa"][1:].view(np.uint8)]), ref_sa)
    F = d["ref_len"] // 2
    assert np.array_equal(d["pac"][:(F + 3) // 4], ref_pac[:(F + 3) // 4])
    gidx.close()


def test_index_build_repeats_and_odd_lengths(gpu_device):
    import ma_amd
    for lens, kw in (([1000, 37, 128, 5000], {}), ([50000, 50001], dict(repeat_unit=700, repeat_copies=60, repeat_div=0.01)),
                     ([64], {}), ([4000], dict(repeat_unit=3000, repeat_copies=3, repeat_div=0.0))):
        g = rand_genome(int(sum(lens)), lens, **kw)
        want = OrIndex.build(g).arrays()
        gidx = ma_amd.Index.build(g)
        got = gidx.download()
        assert got["primary"] == want["primary"] and np.array_equal(got["L2"], want["L2"])
        assert np.array_equal(got["bwt"], want["bwt"]), lens
        assert np.array_equal(got["sa"], want["sa"]), lens
        assert np.array_equal(got["pac"][:len(want["pac"])], want["pac"]), lens
        gidx.close()


@pytest.mark.parametrize("preset", ["default", "illumina"])
def test_pipeline_vs_oracle_heuristics_long_reads(gpu_device, preset):
    import ma_amd
    g = rand_genome(9, [2600000, 1500000, 1000000], repeat_unit=300, repeat_copies=200, repeat_div=0.08)
    reads = (sample_reads(g, 600, 150, 31, sub=0.01) + sample_reads(g, 100, 150, 32, sub=0.06, n_rate=0.01)
             + sample_reads(g, 12, 6000, 33, sub=0.005, ins=0.003, dele=0.003)
             + sample_reads(g, 2, 30000, 34, sub=0.03, ins=0.03, dele=0.04) + sample_reads(g, 20, 150, 35, random_frac=1.0)
             + sample_reads(g, 1, 80000, 36, sub=0.01, ins=0.005, dele=0.005))  # > 48 KB of LDS for the reversed query
    gidx = ma_amd.Index.build(g)
    oidx = OrIndex.from_parts(gidx.download())
    res = oidx.align(reads, or_params(preset, 3), threads=8)
    got, counters, counts = gpu_pipeline(gidx, preset, 3, reads)
    want = []
    for r in range(len(reads)):
        d = dict(len=len(reads[r]))
        d["segs"] = [tuple(int(x) for x in s) for s in res["segs"][int(res["seg_off"][r]):int(res["seg_off"][r + 1])]]
        d["seeds"] = [(int(s["q_start"]), int(s["len"]), int(s["r_start"]), int(s["ambiguity"]), int(s["on_forward"]),
                       int(s["delta"])) for s in res["seeds"][int(res["seed_off"][r]):int(res["seed_off"][r + 1])]]
        d["hsets"] = []
        for h in range(int(res["hset_off"][r]), int(res["hset_off"][r + 1])):
            ss = res["hseeds"][int(res["hseed_off"][h]):int(res["hseed_off"][h + 1])]
            d["hsets"].append(dict(soc=int(res["hset_soc"][h]),
                                   seeds=[(int(s["q_start"]), int(s["len"]), int(s["r_start"]), int(s["ambiguity"]),
                                           int(s["on_forward"]), int(s["delta"])) for s in ss]))
        d["alns"] = []
        for a in res["alns"][int(res["aln_off"][r]):int(res["aln_off"][r + 1])]:
            o = int(a["ops_off"])
            d["alns"].append(dict(bref=int(a["begin_ref"]), eref=int(a["end_ref"]), bq=int(a["begin_q"]),
                                  eq=int(a["end_q"]), score=int(a["score"]), soc=int(a["soc_index"]),
                                  ops=[(int(res["ops"][2 * (o + k)]), int(res["ops"][2 * (o + k) + 1]))
                                       for k in range(int(a["n_ops"]))]))
        want.append(d)
    compare_reads(got, want, what=("segs", "seeds", "hsets", "alns"))
    assert counts["aligned_reads"] == res["n_aligned"]
    gidx.close()


def oracle_reads_as_dicts(res, n):
    want = []
    for r in range(n):
        d = dict()
        d["alns"] = []
        for a in res["alns"][int(res["aln_off"][r]):int(res["aln_off"][r + 1])]:
            o = int(a["ops_off"])
            d["alns"].append(dict(bref=int(a["begin_ref"]), eref=int(a["end_ref"]), bq=int(a["begin_q"]),
                                  eq=int(a["end_q"]), score=int(a["score"]), soc=int(a["soc_index"]),
                                  ops=[(int(res["ops"][2 * (o + k)]), int(res["ops"][2 * (o + k) + 1]))
                                       for k in range(int(a["n_ops"]))]))
        want.append(d)
    return want


@pytest.mark.parametrize("preset", ["default", "illumina"])
def test_extension_early_stop_on_tandem_repeats(gpu_device, preset):
    """The pipeline's extension kernel stops once no later diagonal can raise ez.max (ksw_reg.h).  Microsatellites
    and reads with noisy ends are where a late, higher maximum could appear: alignments must stay identical."""
    import ma_amd
    rng = np.random.default_rng(77)
    contigs = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in (900000, 500000)]
    spots = []
    for c in contigs:
        for _ in range(150):
            period = int(rng.integers(1, 9))
            n = int(rng.integers(40, 900))
            p = int(rng.integers(2000, len(c) - 3000))
            unit = rng.integers(0, 4, size=period, dtype=np.uint8)
            arr = np.tile(unit, n // period + 1)[:n]
            mut = rng.random(n) < 0.03
            arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
            c[p:p + n] = arr
            spots.append((c, p, n))
    reads = []
    for c, p, n in spots:
        for _ in range(4):
            L = int(rng.integers(60, 251))
            start = max(0, int(p + rng.integers(-L, n)))
            r = c[start:start + L].copy()
            k = int(rng.integers(0, 40))  # noisy tail: the extension has to work
            if k and len(r) > k:
                tail = rng.random(k) < 0.35
                seg = r[-k:] if rng.random() < 0.5 else r[:k]
                seg[tail] = (seg[tail] + rng.integers(1, 4, size=int(tail.sum()), dtype=np.uint8)) % 4
            mut = rng.random(len(r)) < 0.02
            r[mut] = (r[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) % 4
            if rng.random() < 0.5:
                r = revcomp(r)
            reads.append(r)
    gidx = ma_amd.Index.build(contigs)
    oidx = OrIndex.from_parts(gidx.download())
    res = oidx.align(reads, or_params(preset, 5), threads=8)
    got, counters, counts = gpu_pipeline(gidx, preset, 5, reads)
    want = oracle_reads_as_dicts(res, len(reads))
    compare_reads(got, want, what=("alns",))
    assert counts["aligned_reads"] == res["n_aligned"]
    assert 0 < int(counters[4]) < int(res["counters"][4])  # the early stop really skipped diagonals
    gidx.close()


def test_index_build_bucketed_round0_path(gpu_device, monkeypatch):
    """GRCh38-sized genomes take the bucketed (first-2-bases) round 0; force it on a small genome."""
    import ma_amd
    g = rand_genome(77, [30000, 999, 20001], repeat_unit=500, repeat_copies=20, repeat_div=0.02)
    want = OrIndex.build(g).arrays()
    for k in ("1", "2"):
        monkeypatch.setenv("MA_INDEX_BUCKET_K", k)
        gidx = ma_amd.Index.build(g)
        got = gidx.download()
        assert got["primary"] == want["primary"]
        assert np.array_equal(got["bwt"], want["bwt"]) and np.array_equal(got["sa"], want["sa"]), k
        gidx.close()
    monkeypatch.delenv("MA_INDEX_BUCKET_K")


def test_full_size_property_reads_map_back_to_origin(gpu_device):
    """Size-independent property at a large scale: on a 300 Mnt synthetic genome (bucketed index build,
    heuristics on) error-free reads must align end-to-end at the position they were sampled from."""
    import ctypes as C
    import torch
    import ma_amd
    L = ma_amd.lib()
    lens = np.array([120000000, 100000000, 80000000], dtype=np.uint64)
    F = int(lens.sum())
    g = torch.empty(F, dtype=torch.uint8, device="cuda")
    assert L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(F), C.c_int32(0), C.c_void_p(g.data_ptr())) == 0
    idx = ma_amd.Index.build_device(lens, g.data_ptr())
    n, rl = 200000, 150
    codes = torch.empty(n * rl + 64, dtype=torch.uint8, device="cuda")
    offs = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    nb = C.c_uint64()
    assert L.ma_synth_reads_device(idx.h, C.c_uint64(11), C.c_uint64(n), C.c_uint32(rl), C.c_double(0.0), C.c_double(0.0),
                                   C.c_double(0.0), C.c_uint64(0), C.c_void_p(codes.data_ptr()),
                                   C.c_void_p(offs.data_ptr()), C.c_uint64(n * rl + 64), C.byref(nb)) == 0
    b = ma_amd.Batch(idx, ma_amd.Params.preset("default"), n, n * rl + 64)
    b.set_reads_device(codes.data_ptr(), offs.data_ptr(), n, int(nb.value))
    b.align()
    b.sync()
    moff, alns, ops = b.mapq_alignments()
    assert b.counts()["aligned_reads"] == n
    first = alns[moff[:-1].astype(np.int64)]
    # error-free 150-mers: one seed, score 2*150, whole query, reference span 150
    assert np.all(first["score"] == 2 * rl)
    assert np.all(first["begin_q"] == 0) and np.all(first["end_q"] == rl)
    assert np.all(first["end_ref"] - first["begin_ref"] == rl)
    # reads come from their true origin: re-extract the reference window and compare with the read
    gh = g.cpu().numpy()
    rc = codes[: n * rl].cpu().numpy().reshape(n, rl)
    for i in range(0, n, 997):
        br = int(first["begin_ref"][i])
        if br < F:
            ref = gh[br:br + rl]
        else:
            ref = 3 - gh[2 * F - (br + rl):2 * F - br][::-1]
        assert np.array_equal(ref, rc[i]), i
    b.close()
    idx.close()
