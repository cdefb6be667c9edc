"""CPU tests of the measurement tooling (round 6): every kernel of the pipeline's stages belongs to a group of
tools/pmc_summarize.py -- a DP kernel family that falls through (round 5: k_ksw_band) is an error, not a silently smaller roofline --
and bench.py's per-kernel roofline block is arithmetic on the committed rocprofv3 rows."""
import csv
import glob
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pmc():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summarize
    importlib.reload(pmc_summarize)
    return pmc_summarize


def test_every_pipeline_kernel_of_the_committed_traces_has_a_group():
    P = _pmc()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[56]_kernel_stats_*.csv")))
    assert files
    seen = set()
    for f in files:
        for row in csv.DictReader(open(f)):
            g = P.group_of(row["Name"])
            if g:
                seen.add(g)
    assert not P.UNMATCHED, P.UNMATCHED
    assert {"k_ksw_band", "k_ksw_ext<1>", "k_ksw_ext<2>", "k_ksw_grp<2>", "k_ksw_grp<4>", "k_ksw_pk", "k_seed"} <= seen
    # the two shapes of the band kernel are families of their own
    assert P.group_of("void ma::k_ksw_band<(anonymous namespace)::PipeFetch, true, 4>((anonymous namespace)::PipeFetch, ma::KswScoring)") == "k_ksw_band"
    assert P.group_of("void ma::k_ksw_band<(anonymous namespace)::PipeFetch, false, 1>((anonymous namespace)::PipeFetch, ma::KswScoring)") == "k_ksw_band (long)"
    assert P.group_of("void k_seed_tasks_smem(TaskKernelArgs)") == "k_seed"


def test_an_unknown_dp_kernel_fails_the_summary(tmp_path):
    d = tmp_path / "prof" / "trace"
    d.mkdir(parents=True)
    with open(d / "x_kernel_stats.csv", "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
        f.write('"void ma::k_ksw_newfamily<int>(int)",4,4000000,1000000,1.0,1,1,0\n')
        f.write('"void k_seed<false>(SeedKernelArgs)",4,8000000,2000000,1.0,1,1,0\n')
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summarize.py"), str(tmp_path / "prof"), "4"], capture_output=True, text=True)
    assert r.returncode == 2 and "k_ksw_newfamily" in r.stderr


def test_per_kernel_roofline_is_the_rows_arithmetic():
    sys.path.insert(0, ROOT)
    import bench
    rows = {"k_ksw_ext<1>": {"ms_per_step": 7.0, "SQ_INSTS_VALU": 3.5e9, "hbm_bytes_per_step": 2.8e9},
            "k_ksw_band (long)": {"ms_per_step": 50.0, "SQ_INSTS_VALU": 2.0e10},
            "k_seed": {"ms_per_step": 8.0, "hbm_bytes_per_step": 1.6e10}, "k_chain": {"ms_per_step": 3.0}}
    fam = [0.0] * 16
    fam[0], fam[1] = 4.0e9, 3.0e5  # k_ksw_ext<1>: cells, jobs per step
    fam[14], fam[15] = 1.0e10, 1.0e5
    pk = bench.per_kernel_roofline(rows, fam)
    assert set(pk) == {"k_ksw_ext<1>", "k_ksw_band (long)", "k_seed"}
    e = pk["k_ksw_ext<1>"]
    assert e["valu_frac"] == pytest.approx(3.5e9 / 7.0e-3 / 1e9 / bench.CHIP_VALU_PEAK_GINST, abs=1e-3)
    assert e["hbm_frac"] == pytest.approx(2.8e9 / 7.0e-3 / 1e9 / bench.HBM_PEAK_GBS, abs=1e-4)
    assert e["lane_insts_per_cell"] == pytest.approx(3.5e9 * 64 / 4.0e9, abs=0.1)
    assert pk["k_ksw_band (long)"]["lane_insts_per_cell"] == pytest.approx(2.0e10 * 64 / 1.0e10, abs=0.1)
    assert "valu_frac" not in pk["k_seed"] and pk["k_seed"]["hbm_frac"] == pytest.approx(1.6e10 / 8e-3 / 1e9 / 8000.0, abs=1e-4)
