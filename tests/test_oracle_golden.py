"""The oracle (oracle/, CPU restatement) against the golden vectors the REAL reference produced
(tests/golden/make_golden.py).  Bit-exact on every integer; mapq compared as printed (%.17g)."""
import filecmp
import os

import numpy as np
import pytest

from ma_testlib import (ROOT, gunzip_to, run_oracle, first_diff, orlib, or_params, OrIndex, read_case,
                        read_ksw_cases, parse_ksw_dump, or_ksw)

G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def small_case(tmp_path_factory):
    d = tmp_path_factory.mktemp("golden")
    return gunzip_to(os.path.join(G, "small.case.gz"), str(d / "small.case")), d


def test_index_files_identical(small_case):
    case, d = small_case
    run_oracle("index", case, str(d / "or"))
    for ext in ("bwt", "sa", "pac"):
        ref = gunzip_to(os.path.join(G, "small_ref.%s.gz" % ext), str(d / ("ref." + ext)))
        assert filecmp.cmp(ref, str(d / ("or." + ext)), shallow=False), ext


def test_extend_backward_traces_identical(small_case):
    case, d = small_case
    run_oracle("ext", case, str(d / "or.ext"))
    ref = gunzip_to(os.path.join(G, "small_ref.ext.gz"), str(d / "ref.ext"))
    assert first_diff(ref, str(d / "or.ext")) is None


@pytest.mark.parametrize("preset,seed,name", [("default", 1, "small_ref.default.pipe"),
                                              ("illumina", 1, "small_ref.illumina.pipe"),
                                              ("default", 7, "small_ref.default.seed7.pipe"),
                                              ("default+mems", 1, "small_ref.mems.pipe")])
def test_pipeline_dump_identical(small_case, preset, seed, name):
    case, d = small_case
    out = str(d / (name + ".or"))
    run_oracle("pipe", case, preset, seed, out)
    ref = gunzip_to(os.path.join(G, name + ".gz"), str(d / name))
    assert first_diff(ref, out) is None


# (match, mismatch, gap, extend, gap2, extend2) of tests/golden/ksw_ref.sc<k>.out.gz (make_golden.py KSW_SCORINGS)
KSW_SCORINGS = [(3, 5, 6, 3, 30, 2), (1, 3, 5, 2, 24, 1), (2, 4, 24, 1, 4, 2), (5, 4, 2, 1, 40, 1)]


def test_ksw_golden(tmp_path):
    case = gunzip_to(os.path.join(G, "ksw.case.gz"), str(tmp_path / "ksw.case"))
    run_oracle("ksw", case, str(tmp_path / "or.out"))
    ref = gunzip_to(os.path.join(G, "ksw_ref.out.gz"), str(tmp_path / "ref.out"))
    assert first_diff(ref, str(tmp_path / "or.out")) is None


@pytest.mark.parametrize("k", range(len(KSW_SCORINGS)))
def test_ksw_golden_other_scorings(tmp_path, k):
    case = gunzip_to(os.path.join(G, "ksw.case.gz"), str(tmp_path / "ksw.case"))
    run_oracle("ksw", case, str(tmp_path / "or.out"), "clean", *KSW_SCORINGS[k])
    ref = gunzip_to(os.path.join(G, "ksw_ref.sc%d.out.gz" % k), str(tmp_path / "ref.out"))
    assert first_diff(ref, str(tmp_path / "or.out")) is None


def test_ksw_edge_cases_via_ctypes():
    p = or_params()
    # empty inputs reset ez only (kswcpp_core.h:362-364)
    ez, cig = or_ksw(p, [], [0, 1, 2], 10, -1, 0)
    assert ez["n_cigar"] == 0 and ez["max"] == 0 and ez["score"] == -2**31 and ez["max_q"] == -1
    # 1x1 match
    ez, cig = or_ksw(p, [2], [2], 10, -1, 0)
    assert ez["score"] == 2 and list(cig) == [1 << 4 | 0]


def test_glibc_rand_restatement_matches_libc():
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    L = orlib()
    L.ma_or_rand.restype = C.c_int32
    for seed in (0, 1, 7, 12345, 2**31 - 1):
        libc.srand(C.c_uint(seed))
        st = (C.c_uint32 * 35)()
        L.ma_or_srand(C.c_uint32(seed), st)
        for _ in range(1000):
            assert libc.rand() == L.ma_or_rand(st)


def test_counters_are_consistent(small_case):
    case, d = small_case
    contigs, reads, _ = read_case(case)
    idx = OrIndex.build(contigs)
    res = idx.align(reads[:40], or_params(), threads=2)
    c = res["counters"]
    assert c[0] > 0 and c[0] <= c[1] <= 2 * c[0]  # blocks per extend_backward in [1,2]
    assert c[3] == len(res["seeds"])
    assert c[5] > 0 and c[4] > 0


# (preset, search inversions, paired, Z Drop Inversions, SAM options) of tests/golden/f4.* (make_golden.py F4_CONFIGS)
F4_CONFIGS = [("default", 1, 0, 100, 0), ("default", 1, 1, 100, 0), ("illumina", 0, 1, 100, 3), ("default", 1, 1, 40, 1)]


@pytest.mark.parametrize("cfg", F4_CONFIGS)
def test_f4_golden(tmp_path, cfg):
    """SmallInversions + PairedReads of the oracle against the reference's lists."""
    preset, inv, paired, zd, opt = cfg
    nm = "f4.%s.inv%d.pair%d.zd%d.opt%d" % cfg
    case = gunzip_to(os.path.join(G, "f4.case.gz"), str(tmp_path / "f4.case"))
    run_oracle("f4", case, preset, 1, str(tmp_path / "or.f4"), inv, paired, zd)
    ref = gunzip_to(os.path.join(G, nm + ".f4.gz"), str(tmp_path / "ref.f4"))
    assert first_diff(ref, str(tmp_path / "or.f4")) is None


def test_oracle_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The checker itself under clang's -fsanitize=address,undefined: the index files, the extend_backward traces, the pipeline dumps of
    three parameter sets and the kswcpp cases of the compiled reference come out byte for byte, with no sanitizer report -- an
    oracle that agrees with the reference through undefined behaviour would be no oracle."""
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang")
    exe = str(tmp_path / "oracle_dump_san")
    subprocess.check_call([clang, "-std=c++17", "-O1", "-g", "-msse4.1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-w",
                           os.path.join(ROOT, "oracle", "ma_oracle.cpp"), os.path.join(ROOT, "oracle", "oracle_dump.cpp"), "-o", exe, "-lpthread"])

    def run(*args):
        p = subprocess.run([exe] + [str(a) for a in args], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "runtime error" not in p.stderr and "Sanitizer" not in p.stderr, p.stderr[-2000:]

    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    run("index", case, str(tmp_path / "or"))
    for ext in ("bwt", "sa", "pac"):
        assert filecmp.cmp(gunzip_to(os.path.join(G, "small_ref.%s.gz" % ext), str(tmp_path / ("ref." + ext))), str(tmp_path / ("or." + ext)), shallow=False), ext
    run("ext", case, str(tmp_path / "or.ext"))
    assert first_diff(gunzip_to(os.path.join(G, "small_ref.ext.gz"), str(tmp_path / "ref.ext")), str(tmp_path / "or.ext")) is None
    for preset, seed, name in (("default", 1, "small_ref.default.pipe"), ("illumina", 1, "small_ref.illumina.pipe"), ("default+mems", 1, "small_ref.mems.pipe")):
        run("pipe", case, preset, seed, str(tmp_path / "or.pipe"))
        assert first_diff(gunzip_to(os.path.join(G, name + ".gz"), str(tmp_path / name)), str(tmp_path / "or.pipe")) is None, name
    kcase = gunzip_to(os.path.join(G, "ksw.case.gz"), str(tmp_path / "ksw.case"))
    run("ksw", kcase, str(tmp_path / "or.out"))
    assert first_diff(gunzip_to(os.path.join(G, "ksw_ref.out.gz"), str(tmp_path / "ref.out")), str(tmp_path / "or.out")) is None
