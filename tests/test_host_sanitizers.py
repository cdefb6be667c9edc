"""The host layer of the drop-in (ma_amd/host/: SAM writer, FASTA/FASTQ readers, PairedReads, SmallInversions' bookkeeping -- SURVEY.md
section 8 rows f3 / f4) under clang's -fsanitize=address,undefined: the drivers of tests/test_sam_writer.py and tests/test_f4_host.py
write the reference's golden bytes with no sanitizer report.  (The kernels' stage logic and the oracle: tests/test_host_logic.py,
tests/test_oracle_golden.py; the prefetch queue under ThreadSanitizer: tests/test_prefetch_queue.py.)"""
import gzip
import os
import subprocess

import pytest

from ma_testlib import ROOT, gunzip_to

G = os.path.join(ROOT, "tests", "golden")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _build(tmp_path, name):
    if not os.path.exists(CLANG):
        pytest.skip("no clang")
    if not os.path.exists(os.path.join(ROOT, "ma_amd", "libma_amd.so")):
        pytest.skip("libma_amd.so is not built")
    exe = str(tmp_path / (name + "_san"))
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-w", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "ma_amd", "host"), os.path.join(ROOT, "tests", "emul", name + ".cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "ma_amd"), "-lma_amd", "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"), "-lpthread"])
    return exe


def _run(args, cwd=None):
    p = subprocess.run([str(a) for a in args], cwd=cwd, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "runtime error" not in p.stderr and "Sanitizer" not in p.stderr, p.stderr[-2000:]


def test_sam_writer_and_readers_under_sanitizers(tmp_path):
    sam = _build(tmp_path, "sam_test")
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    for preset, opts in (("default", (0, 3, 5, 8 | 1)), ("illumina", (0, 4))):
        pipe = gunzip_to(os.path.join(G, "small_ref.%s.pipe.gz" % preset), str(tmp_path / ("p.%s.pipe" % preset)))
        for opt in opts:
            out = str(tmp_path / "o.sam")
            _run([sam, case, pipe, out, opt])
            want = gzip.open(os.path.join(G, "small_ref.%s.opt%d.sam.gz" % (preset, opt & 7)), "rt").read()
            assert open(out).read() == want, (preset, opt)
    reader = _build(tmp_path, "reader_test")
    for rc in (0, 1):
        out = str(tmp_path / "o.txt")
        _run([reader, os.path.join("reader", "mates_1.fq"), out, os.path.join("reader", "mates_2.fq"), rc], cwd=G)
        assert open(out).read() == open(os.path.join(G, "reader", "mates.rc%d.ref" % rc)).read()


def test_pairing_and_inversion_records_under_sanitizers(tmp_path):
    f4 = _build(tmp_path, "f4_test")
    case = gunzip_to(os.path.join(G, "f4.case.gz"), str(tmp_path / "f4.case"))
    for cfg in (("default", 1, 1, 100, 0), ("illumina", 0, 1, 100, 3), ("default", 1, 0, 100, 0), ("default", 1, 1, 40, 1)):
        nm = "f4.%s.inv%d.pair%d.zd%d.opt%d" % cfg
        dump = gunzip_to(os.path.join(G, nm + ".f4.gz"), str(tmp_path / "ref.f4"))
        _run([f4, case, dump, cfg[0], cfg[2], cfg[4], tmp_path / "o.f4", tmp_path / "o.sam"])
        assert open(str(tmp_path / "o.f4")).read() == gzip.open(os.path.join(G, nm + ".f4.gz"), "rt").read(), nm
        assert open(str(tmp_path / "o.sam")).read() == gzip.open(os.path.join(G, nm + ".sam.gz"), "rt").read(), nm
