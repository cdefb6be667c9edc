"""Product stage logic (ma_amd/csrc/{seeding,chain,nw,stdsort,fm_device}.h) compiled for the CPU by
tests/emul/host_emul.cpp and diffed against the oracle and the reference's golden dumps.  The shipped
path runs the same functions inside HIP kernels; this is the 'host logic' part of the CPU suite."""
import os
import subprocess

import pytest

from ma_testlib import ROOT, build_oracle, gunzip_to, first_diff, rand_genome, sample_reads, write_case, run_oracle

EMUL = os.path.join(ROOT, "tests", "emul", "host_emul")
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def emul():
    build_oracle()
    src = os.path.join(ROOT, "tests", "emul", "host_emul.cpp")
    deps = [src] + [os.path.join(ROOT, "ma_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "ma_amd", "csrc"))
                    if f.endswith(".h")]
    if not os.path.exists(EMUL) or any(os.path.getmtime(d) > os.path.getmtime(EMUL) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-w", "-I" + os.path.join(ROOT, "include"),
                               src, "-o", EMUL, "-L" + os.path.join(ROOT, "oracle"), "-lma_oracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return EMUL


def test_round3_units(emul):
    """K-mer keys by bit gathering == the byte loops (seeding.h, seed_center); the window sweep with prefix sums == the sweep
    with running sums and re-added strips (chain.h, soc_windows)."""
    out = subprocess.check_output([emul, "x", "default", "7", "/dev/null", "round3check"]).decode()
    assert "round3check ok" in out


def test_stdsort_matches_libstdcxx(emul):
    out = subprocess.check_output([emul, "x", "default", "3", "/dev/null", "sortcheck"]).decode()
    assert "sortcheck ok" in out


@pytest.mark.parametrize("preset,name", [("default", "small_ref.default.pipe"), ("illumina", "small_ref.illumina.pipe"),
                                         ("default+mems", "small_ref.mems.pipe")])
def test_stage_logic_vs_reference_golden(emul, tmp_path, preset, name):
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    ref = gunzip_to(os.path.join(G, name + ".gz"), str(tmp_path / name))
    out = str(tmp_path / "emul.pipe")
    subprocess.check_call([emul, case, preset, "1", out, "all"])
    assert first_diff(ref, out) is None


def test_smem_twin_merge_vs_reference_golden(emul, tmp_path):
    """The SMEM state machine with the kernels' default (list entries whose interval equals that of the entry before them are
    not kept, seeding.h seed_apply) reproduces the compiled reference's dump of the Illumina preset."""
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    ref = gunzip_to(os.path.join(G, "small_ref.illumina.pipe.gz"), str(tmp_path / "ref.pipe"))
    out = str(tmp_path / "emul.pipe")
    subprocess.check_call([emul, case, "illumina", "1", out, "all"], env=dict(os.environ, MA_EMUL_SMEM_MERGE="1"))
    assert first_diff(ref, out) is None


def test_smem_compact_entries_vs_reference_golden_and_long_reads(emul, tmp_path):
    """The 16-byte SMEM list entries (seeding.h smem_pack: three 35-bit interval fields + ONE 22-bit length; the start of the
    match is shared by all entries of a list and named by the reader) give the compiled reference's Illumina dump and, for reads
    of 2047 / 2048 / 20 000 bases (round 4's two 11-bit fields ended at 2047), the oracle's.  The 40-byte form of the same run
    aborts if an entry's own start ever differs from the shared one (smem_get, host build).  Round 6: the same with the heads of the
    two lists in a separate array that starts out as garbage (SeedScratch::lds -- LDS in k_seed_tasks_smem)."""
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    ref = gunzip_to(os.path.join(G, "small_ref.illumina.pipe.gz"), str(tmp_path / "ref.pipe"))
    out = str(tmp_path / "emul.pipe")
    for merge in ("0", "1"):
        subprocess.check_call([emul, case, "illumina", "1", out, "all"],
                              env=dict(os.environ, MA_EMUL_SMEM_MERGE=merge, MA_EMUL_SMEM_COMPACT="1"))
        assert first_diff(ref, out) is None
        # round 6: the first 1 / 3 / 6 entries of each list in their own array (k_seed_tasks_smem keeps them in LDS)
        for heads in ("1", "3", "6"):
            subprocess.check_call([emul, case, "illumina", "1", out, "all"],
                                  env=dict(os.environ, MA_EMUL_SMEM_MERGE=merge, MA_EMUL_SMEM_COMPACT="1", MA_EMUL_SMEM_LDS_HEADS=heads))
            assert first_diff(ref, out) is None, (merge, heads)
    g = rand_genome(78, [300000, 200000], repeat_unit=250, repeat_copies=60, repeat_div=0.05)
    reads = (sample_reads(g, 4, 2047, 1, sub=0.03, ins=0.02, dele=0.02) + sample_reads(g, 4, 2048, 2, sub=0.03, ins=0.02, dele=0.02)
             + sample_reads(g, 1, 20000, 3, sub=0.03, ins=0.03, dele=0.03, n_rate=0.001))
    c2 = str(tmp_path / "c.case")
    write_case(c2, g, reads)
    run_oracle("pipe", c2, "illumina", 5, str(tmp_path / "or.pipe"))
    for compact, heads in (("0", "0"), ("1", "0"), ("1", "6"), ("1", "2")):
        subprocess.check_call([emul, c2, "illumina", "5", str(tmp_path / "em.pipe"), "all"],
                              env=dict(os.environ, MA_EMUL_SMEM_MERGE="1", MA_EMUL_SMEM_COMPACT=compact, MA_EMUL_SMEM_LDS_HEADS=heads))
        assert first_diff(str(tmp_path / "or.pipe"), str(tmp_path / "em.pipe")) is None, (compact, heads)


def test_stage_logic_vs_oracle_long_reads(emul, tmp_path):
    g = rand_genome(77, [400000, 250000], repeat_unit=250, repeat_copies=60, repeat_div=0.05)
    reads = (sample_reads(g, 120, 150, 1, sub=0.02) + sample_reads(g, 6, 5000, 2, sub=0.01, ins=0.005, dele=0.005)
             + sample_reads(g, 1, 20000, 3, sub=0.03, ins=0.03, dele=0.03))
    case = str(tmp_path / "c.case")
    write_case(case, g, reads)
    for preset in ("default", "illumina", "default+mems"):
        run_oracle("pipe", case, preset, 5, str(tmp_path / "or.pipe"))
        subprocess.check_call([emul, case, preset, "5", str(tmp_path / "em.pipe"), "all"])
        assert first_diff(str(tmp_path / "or.pipe"), str(tmp_path / "em.pipe")) is None


def test_window_sweep_restated_for_one_wavefront_is_the_references_loop():
    """k_soc_windows_wave (stage_chain.h) restates the SoC sweep (stripOfConsideration.cpp:77-113 with push_back_no_overlap,
    soc.h:362-404) as prefix sums + window ends by binary search + a strip stack whose top two entries carry the prefix sums at
    their ends.  Both forms in plain Python on random seed lists with heavy ties in delta, several contigs and every strip
    width: the same strips in the same order.  The binary search needs contig ids that do not fall along the deltas; lists
    where they do are the ones the kernel hands to its lane form (and the restatement must then NOT be trusted: counted)."""
    import numpy as np
    rng = np.random.default_rng(17)

    def less(a, b):  # SoCOrder::operator< (soc.h:71-76)
        return a[1] > b[1] if a[0] == b[0] else a[0] < b[0]

    def push(mx, cur, itS, itE, minScore, resum):
        cur = list(cur)
        while mx and mx[-1][3] > itS:
            back = mx[-1]
            if less(back, cur):
                bb = back[2]
                back[0], back[1] = resum(bb, itS)
                back[3] = itS
                if back[0] < minScore or back[0] == 0:
                    mx.pop()
            else:
                be = back[3]
                cur[0], cur[1] = resum(be, itE)
                itS = be
                if cur[0] < minScore or cur[0] == 0:
                    return
        mx.append([cur[0], cur[1], itS, itE])

    def reference_form(delta, cid, ln, amb, strip, fmin):
        n = len(delta)
        resum = lambda b, e: (int(ln[b:e].sum()), int(amb[b:e].sum())) if e >= b else (0, 0)
        mx, S, E, acc, am = [], 0, 0, 0, 0
        while E != n and S != n:
            while E != n and delta[S] + strip >= delta[E] and cid[S] == cid[E]:
                acc += int(ln[E]); am += int(amb[E]); E += 1
            if acc >= fmin:
                push(mx, (acc, am), S, E, int(fmin), resum)
            acc -= int(ln[S]); am -= int(amb[S]); S += 1
        return mx

    def wave_form(delta, cid, ln, amb, strip, fmin):
        n = len(delta)
        pl = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        pa = np.concatenate([[0], np.cumsum(amb)]).astype(np.int64)
        winE = np.zeros(n, dtype=np.int64)
        for S in range(n):
            lo, hi = S + 1, n
            while lo < hi:
                mid = lo + (hi - lo) // 2
                if delta[mid] > delta[S] + strip or cid[mid] != cid[S]:
                    hi = mid
                else:
                    lo = mid + 1
            winE[S] = lo
        # the stack as the kernel keeps it: the array in memory (written whenever an entry changes and stays), the top two entries
        # in registers with the prefix sums at their ends -- [accLen, amb, b, e, plb, pab, ple, pae]
        mem, nmx, top, second = {}, 0, None, None
        minScore = int(fmin)
        for S in range(n):
            E = int(winE[S])
            plS, paS, plE, paE = int(pl[S]), int(pa[S]), int(pl[E]), int(pa[E])
            cur = [plE - plS, paE - paS]
            if cur[0] >= fmin:
                itS, plI, paI, drop = S, plS, paS, False
                while nmx > 0 and top[3] > itS:
                    if less(top, cur):
                        top[0], top[1] = (0, 0) if itS < top[2] else (plI - top[4], paI - top[5])
                        top[3], top[6], top[7] = itS, plI, paI
                        if top[0] < minScore or top[0] == 0:
                            nmx -= 1
                            if nmx > 0:
                                if second is not None:
                                    top, second = second, None
                                else:
                                    m = mem[nmx - 1]
                                    top = m + [int(pl[m[2]]), int(pa[m[2]]), int(pl[m[3]]), int(pa[m[3]])]
                        else:
                            mem[nmx - 1] = top[:4]
                    else:
                        be = top[3]
                        cur = [0, 0] if E < be else [plE - top[6], paE - top[7]]
                        itS, plI, paI = be, top[6], top[7]
                        if cur[0] < minScore or cur[0] == 0:
                            drop = True
                            break
                if not drop:
                    if nmx > 0:
                        second = list(top)
                    top = [cur[0], cur[1], itS, E, plI, paI, plE, paE]
                    mem[nmx] = top[:4]
                    nmx += 1
            if E == n:
                break
        return [mem[k] for k in range(nmx)]

    falling = 0
    for case in range(1500):
        n = int(rng.integers(1, 120))
        n_contigs = int(rng.integers(1, 4))
        cid = np.sort(rng.integers(0, n_contigs, size=n))
        delta = np.sort(rng.integers(0, int(rng.choice([8, 60, 2000])), size=n)).astype(np.int64)
        if case % 5 == 4 and n > 3:  # contig ids that fall along the deltas somewhere
            i = int(rng.integers(0, n - 1))
            cid[i], cid[i + 1] = max(cid[i], cid[i + 1]) + 1, min(cid[i], cid[i + 1])
        ln = rng.integers(1, 40, size=n).astype(np.int64)
        amb = rng.integers(1, 4, size=n).astype(np.int64)
        strip = int(rng.choice([0, 3, 25, 500]))
        fmin = float(rng.choice([0.0, 16.0, 60.5, 300.0]))
        monotone = bool(np.all(np.diff(cid) >= 0))
        a = reference_form(delta, cid, ln, amb, strip, fmin)
        if not monotone:
            falling += 1
            continue
        b = wave_form(delta, cid, ln, amb, strip, fmin)
        assert a == b, (case, n, strip, fmin)
    assert 100 < falling < 400


def test_stage_logic_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """GPU AddressSanitizer is not available on the pool: the sanitizers run on the CPU build of the SAME stage logic
    (tests/emul/host_emul.cpp over ma_amd/csrc/{seeding,chain,nw,stdsort,fm_device}.h), compiled with clang's
    -fsanitize=address,undefined.  The reference's golden dumps (Default and Illumina presets; SMEM lists with their heads in a
    separate array) and long reads against the oracle: no sanitizer report, the same bytes.  (g++'s -fsanitize=shift is not
    used: it miscomputes the index of `(c ? a : b)[k >> s]` in fm_device.h's sa_sample -- the shift exponent it checks is a
    pointer's low word -- and changes the program's results without any report.)"""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang")
    build_oracle()
    exe = str(tmp_path / "host_emul_san")
    subprocess.check_call([clang, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-w",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "emul", "host_emul.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "oracle"), "-lma_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])

    def run(case, preset, seed, out, **env):
        p = subprocess.run([exe, case, preset, str(seed), out, "all"], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "runtime error" not in p.stderr and "Sanitizer" not in p.stderr, p.stderr[-2000:]

    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "emul.pipe")
    run(case, "default", 1, out)
    assert first_diff(gunzip_to(os.path.join(G, "small_ref.default.pipe.gz"), str(tmp_path / "ref_d.pipe")), out) is None
    ref = gunzip_to(os.path.join(G, "small_ref.illumina.pipe.gz"), str(tmp_path / "ref_i.pipe"))
    for env in (dict(), dict(MA_EMUL_SMEM_MERGE="1", MA_EMUL_SMEM_COMPACT="1", MA_EMUL_SMEM_LDS_HEADS="6")):
        run(case, "illumina", 1, out, **env)
        assert first_diff(ref, out) is None, env
    g = rand_genome(78, [300000, 200000], repeat_unit=250, repeat_copies=60, repeat_div=0.05)
    reads = (sample_reads(g, 4, 2047, 1, sub=0.03, ins=0.02, dele=0.02) + sample_reads(g, 2, 20000, 3, sub=0.03, ins=0.03, dele=0.03, n_rate=0.001)
             + sample_reads(g, 30, 150, 4) + sample_reads(g, 6, 6000, 5, sub=0.01, ins=0.005, dele=0.005))
    c2 = str(tmp_path / "long.case")
    write_case(c2, g, reads)
    for preset, env in (("default", dict()), ("illumina", dict(MA_EMUL_SMEM_MERGE="1", MA_EMUL_SMEM_COMPACT="1", MA_EMUL_SMEM_LDS_HEADS="4"))):
        run_oracle("pipe", c2, preset, 5, str(tmp_path / "or.pipe"))
        run(c2, preset, 5, out, **env)
        assert first_diff(str(tmp_path / "or.pipe"), out) is None, preset
