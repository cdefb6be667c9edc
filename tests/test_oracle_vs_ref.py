"""Differential tests of the oracle against the real reference compiled into oracle/_ref (only where
/root/reference was available to build it; skipped otherwise -- the committed golden vectors in
tests/golden cover that case)."""
import os

import pytest

from ma_testlib import (sample_inversion_reads, sample_pairs, have_ref, rand_genome, sample_reads, write_case, write_ksw_cases, rand_ksw_cases, run_ref,
                        run_oracle, first_diff)

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (no /root/reference here)")


def test_ksw_random_cases(tmp_path):
    cases = rand_ksw_cases(4000, 991, max_len=150) + rand_ksw_cases(40, 992, long_frac=1.0)
    p = str(tmp_path / "k.case")
    write_ksw_cases(p, cases)
    run_ref("ksw", p, str(tmp_path / "ref.out"))
    run_oracle("ksw", p, str(tmp_path / "or.out"))
    assert first_diff(str(tmp_path / "ref.out"), str(tmp_path / "or.out")) is None


# match, mismatch, gap, extend, gap2, extend2 -- the third has q2 + e2 < q + e, which makes
# kswcpp swap the two models internally (kswcpp_core.h:367-377) while still seeding H[0] from the un-swapped one
OTHER_SCORINGS = [(3, 5, 6, 3, 30, 2), (1, 3, 5, 2, 24, 1), (2, 4, 24, 1, 4, 2), (5, 4, 2, 1, 40, 1)]


@pytest.mark.parametrize("sc", OTHER_SCORINGS)
def test_ksw_other_scoring_schemes(tmp_path, sc):
    cases = rand_ksw_cases(1500, 993 + sc[0], max_len=150) + rand_ksw_cases(12, 994, long_frac=1.0)
    p = str(tmp_path / "k.case")
    write_ksw_cases(p, cases)
    run_ref("ksw", p, str(tmp_path / "ref.out"), "clean", *sc)
    run_oracle("ksw", p, str(tmp_path / "or.out"), "clean", *sc)
    assert first_diff(str(tmp_path / "ref.out"), str(tmp_path / "or.out")) is None


# No whole-pipeline test under other scoring schemes: `pGlobalParams` is a namespace-scope const shared_ptr defined in a
# header (parameter.h:1064), i.e. every translation unit of the reference holds its OWN copy; a value set by the caller
# reaches the code inlined into the caller's unit (NeedlemanWunsch's kswcpp parameters) but not alignment.cpp or
# harmonization.cpp, so the reference's pipeline under non-default scoring is not a single well-defined function.
# kswcpp itself takes its parameters explicitly and is pinned above.


@pytest.mark.parametrize("preset", ["default", "illumina", "default+mems", "illumina+mems"])
def test_pipeline_with_heuristics_and_repeats(tmp_path, preset):
    # doubled length > 10 Mnt so that the genome-size gated heuristics are active
    g = rand_genome(5, [2600000, 1500000, 1000000], repeat_unit=300, repeat_copies=200, repeat_div=0.08)
    reads = (sample_reads(g, 250, 150, 31, sub=0.01) + sample_reads(g, 60, 150, 32, sub=0.06, n_rate=0.01)
             + sample_reads(g, 8, 6000, 33, sub=0.005, ins=0.003, dele=0.003)
             + sample_reads(g, 2, 30000, 34, sub=0.03, ins=0.03, dele=0.04) + sample_reads(g, 10, 150, 35, random_frac=1.0))
    p = str(tmp_path / "c.case")
    write_case(p, g, reads)
    run_ref("pipe", p, preset, 3, str(tmp_path / "ref.pipe"))
    run_oracle("pipe", p, preset, 3, str(tmp_path / "or.pipe"))
    assert first_diff(str(tmp_path / "ref.pipe"), str(tmp_path / "or.pipe")) is None


@pytest.mark.parametrize("preset,inv,paired,zdrop_inv", [("default", 1, 0, 100), ("default", 1, 0, 30), ("default", 0, 1, 100),
                                                         ("illumina", 1, 1, 100), ("default", 1, 1, 60)])
def test_f4_small_inversions_and_paired_reads(tmp_path, preset, inv, paired, zdrop_inv):
    """SmallInversions (smallInversions.h) and PairedReads (pairedReads.cpp) after MappingQuality."""
    g = rand_genome(21, [60000, 35000, 12000], repeat_unit=250, repeat_copies=30, repeat_div=0.05)
    reads = (sample_pairs(g, 250, 150, 71) + sample_pairs(g, 30, 250, 72, sub=0.05, insert_mean=600, insert_std=200)
             + sample_inversion_reads(g, 24, 1400, 73) + sample_inversion_reads(g, 6, 4000, 74, inv_min=300, inv_max=900, sub=0.03))
    p = str(tmp_path / "c.case")
    write_case(p, g, reads)
    run_ref("f4", p, preset, 5, str(tmp_path / "ref.f4"), inv, paired, zdrop_inv)
    run_oracle("f4", p, preset, 5, str(tmp_path / "or.f4"), inv, paired, zdrop_inv)
    assert first_diff(str(tmp_path / "ref.f4"), str(tmp_path / "or.f4")) is None
    if inv:  # the case must actually contain inversions that are found
        with open(str(tmp_path / "ref.f4")) as f:
            assert sum(1 for l in f if l.startswith("f ") and " 0:" not in l) >= 10


def test_index_and_traces(tmp_path):
    import filecmp
    g = rand_genome(8, [70000, 1, 129, 40000])
    reads = sample_reads(g[:1], 80, 120, 41, sub=0.02)
    p = str(tmp_path / "c.case")
    write_case(p, g, reads)
    run_ref("index", p, str(tmp_path / "ref"))
    run_oracle("index", p, str(tmp_path / "or"))
    for ext in ("bwt", "sa", "pac"):
        assert filecmp.cmp(str(tmp_path / ("ref." + ext)), str(tmp_path / ("or." + ext)), shallow=False)
    run_ref("ext", p, str(tmp_path / "ref.ext"))
    run_oracle("ext", p, str(tmp_path / "or.ext"))
    assert first_diff(str(tmp_path / "ref.ext"), str(tmp_path / "or.ext")) is None
