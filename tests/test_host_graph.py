"""The drop-in boundary on the host side: ma_amd/host/{ms_graph.h,ma_modules.h} wired exactly like
libMA::setUpCompGraph (export.cpp:99-126) by tests/emul/host_graph_test.cpp."""
import os
import subprocess

import pytest

import gzip

from ma_testlib import ROOT, gunzip_to, parse_pipe_dump

EXE = os.path.join(ROOT, "tests", "emul", "host_graph_test")
G = os.path.join(ROOT, "tests", "golden")


def build_exe():
    src = os.path.join(ROOT, "tests", "emul", "host_graph_test.cpp")
    deps = [src, os.path.join(ROOT, "ma_amd", "host", "ms_graph.h"), os.path.join(ROOT, "ma_amd", "host", "ma_modules.h"),
            os.path.join(ROOT, "ma_amd", "host", "ma_sam.h"), os.path.join(ROOT, "include", "ma_amd.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ma_amd", "host"), src, "-o", EXE,
                               "-L" + os.path.join(ROOT, "ma_amd"), "-lma_amd",
                               "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"), "-lpthread"])
    return EXE


def test_modules_compile_and_fail_loudly_without_gpu(tmp_path):
    exe = build_exe()
    try:
        import ma_amd
        n = ma_amd.device_count()
    except Exception:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present: the no-GPU error path cannot be exercised here")
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = subprocess.check_output([exe, case, "default", str(tmp_path / "o"), "nogpu"]).decode()
    assert "std::runtime_error" in out


@pytest.mark.gpu
@pytest.mark.parametrize("preset,name", [("default", "small_ref.default.pipe"), ("illumina", "small_ref.illumina.pipe")])
def test_graph_of_dropin_modules_matches_reference(tmp_path, gpu_device, preset, name):
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "graph.out")
    subprocess.check_call([exe, case, preset, out])
    got = parse_pipe_dump(out)
    want = parse_pipe_dump(os.path.join(G, name + ".gz"))
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g["alns"] == w["alns"], "read %d alignments" % i
        assert len(g["mq"]) == len(w["mq"])
        for a, b in zip(g["mq"], w["mq"]):
            assert a == b, "read %d mapq record" % i
    # the FileWriter node of the same graph: SAM text identical to the reference's
    sam_want = gzip.open(os.path.join(G, "small_ref.%s.opt0.sam.gz" % preset), "rt").read()
    assert open(out + ".sam").read() == sam_want
