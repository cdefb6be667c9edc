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
    aborts if an entry's own start ever differs from the shared one (smem_get, host build)."""
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    ref = gunzip_to(os.path.join(G, "small_ref.illumina.pipe.gz"), str(tmp_path / "ref.pipe"))
    out = str(tmp_path / "emul.pipe")
    for merge in ("0", "1"):
        subprocess.check_call([emul, case, "illumina", "1", out, "all"],
                              env=dict(os.environ, MA_EMUL_SMEM_MERGE=merge, MA_EMUL_SMEM_COMPACT="1"))
        assert first_diff(ref, out) is None
    g = rand_genome(78, [300000, 200000], repeat_unit=250, repeat_copies=60, repeat_div=0.05)
    reads = (sample_reads(g, 4, 2047, 1, sub=0.03, ins=0.02, dele=0.02) + sample_reads(g, 4, 2048, 2, sub=0.03, ins=0.02, dele=0.02)
             + sample_reads(g, 1, 20000, 3, sub=0.03, ins=0.03, dele=0.03, n_rate=0.001))
    c2 = str(tmp_path / "c.case")
    write_case(c2, g, reads)
    run_oracle("pipe", c2, "illumina", 5, str(tmp_path / "or.pipe"))
    for compact in ("0", "1"):
        subprocess.check_call([emul, c2, "illumina", "5", str(tmp_path / "em.pipe"), "all"],
                              env=dict(os.environ, MA_EMUL_SMEM_MERGE="1", MA_EMUL_SMEM_COMPACT=compact))
        assert first_diff(str(tmp_path / "or.pipe"), str(tmp_path / "em.pipe")) is None


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
