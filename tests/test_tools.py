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


def test_seeding_task_kernels_keep_their_lane_state_out_of_scratch_memory(tmp_path):
    """Round 6: the two selects of seed_qbyte (seeding.h) had been folded into one load through a selected address, which kept the whole
    SeedLane of k_seed_tasks / k_seed_tasks_smem in scratch memory -- 83 / 90 scratch stores in their loops, 0.76 TB of writes per
    Nanopore step, invisible in the spill counts.  The gfx950 assembly of the shipped sources (hipcc cross-compiles without a GPU) must
    have no scratch traffic in those kernels beyond the handful of SGPR-spill slots, and no VGPR spills in any seeding kernel."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = str(tmp_path / "pipeline.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "pipeline.hip", "-o", asm],
                          cwd=os.path.join(ROOT, "ma_amd", "csrc"), stderr=subprocess.DEVNULL)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "kernel_regs.py"), asm, "k_seed"], text=True)
    rows = {}
    for line in out.splitlines():
        t = line.split()
        rows[t[0]] = dict(vgpr=int(t[2]), vspill=int(t[4]), scratch=int(t[10]), lds=int(t[12]))
    tasks = [k for k in rows if "k_seed_tasks" in k]
    assert len(tasks) == 2, rows.keys()
    for k, r in rows.items():
        if "k_seed_rows" in k or "k_seed_final" in k:
            continue
        assert r["vspill"] == 0, (k, r)
    text = open(asm).read()
    for k in tasks:
        body = text[text.index("\n" + k + ":"):]
        body = body[:body.index("s_endpgm")]
        assert body.count("scratch_store") <= 8 and body.count("scratch_load") <= 8, (k, body.count("scratch_store"), body.count("scratch_load"))
        assert rows[k]["scratch"] <= 96, (k, rows[k])
    smem = [k for k in tasks if "smem" in k][0]
    assert rows[smem]["scratch"] == 0 and rows[smem]["lds"] == 49152, rows[smem]  # no scratch at all; six heads of two lists per lane in LDS
