"""SAM emission of the host layer (ma_amd/host/ma_sam.h) against the text the reference's FileWriter
printed for the same reads (tests/golden/small_ref.*.sam.gz, made by make_golden.py from oracle/_ref)."""
import gzip
import os
import subprocess

import pytest

from ma_testlib import ROOT, gunzip_to

G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "tests", "emul", "sam_test")


def build_exe():
    src = os.path.join(ROOT, "tests", "emul", "sam_test.cpp")
    deps = [src] + [os.path.join(ROOT, "ma_amd", "host", h) for h in ("ma_sam.h", "ma_modules.h", "ms_graph.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ma_amd", "host"), src, "-o", EXE,
                               "-L" + os.path.join(ROOT, "ma_amd"), "-lma_amd", "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"),
                               "-lpthread"])
    return EXE


@pytest.mark.parametrize("preset,opt", [("default", 0), ("default", 1), ("default", 2), ("default", 3), ("illumina", 0),
                                        ("default", 4), ("default", 5), ("illumina", 4)])  # bit 2: NGMLR tag emulation
def test_sam_text_matches_reference_filewriter(tmp_path, preset, opt):
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    pipe = gunzip_to(os.path.join(G, "small_ref.%s.pipe.gz" % preset), str(tmp_path / "p.pipe"))
    out = str(tmp_path / "o.sam")
    subprocess.check_call([exe, case, pipe, out, str(opt)])
    want = gzip.open(os.path.join(G, "small_ref.%s.opt%d.sam.gz" % (preset, opt)), "rt").read()
    got = open(out).read()
    assert got.count("\n") == want.count("\n")
    for i, (a, b) in enumerate(zip(got.split("\n"), want.split("\n"))):
        assert a == b, "SAM line %d differs" % i


def test_buffered_writer_writes_the_same_bytes(tmp_path):
    """FileWriter::uiBufferBytes > 0 (per-thread buffers, the lock once per 4 KB instead of once per read): after flush( ) the
    stream holds the bytes of the reference's FileWriter (one thread: even the order)."""
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    pipe = gunzip_to(os.path.join(G, "small_ref.default.pipe.gz"), str(tmp_path / "p.pipe"))
    out = str(tmp_path / "o.sam")
    subprocess.check_call([exe, case, pipe, out, str(8 | 1)])
    assert open(out).read() == gzip.open(os.path.join(G, "small_ref.default.opt1.sam.gz"), "rt").read()


READER_EXE = os.path.join(ROOT, "tests", "emul", "reader_test")


def build_reader_exe():
    src = os.path.join(ROOT, "tests", "emul", "reader_test.cpp")
    deps = [src] + [os.path.join(ROOT, "ma_amd", "host", h) for h in ("ma_sam.h", "ma_modules.h", "ms_graph.h")]
    if not os.path.exists(READER_EXE) or any(os.path.getmtime(d) > os.path.getmtime(READER_EXE) for d in deps):
        zl = ["-DMA_WITH_ZLIB"] if os.path.exists("/usr/include/zlib.h") else []
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall"] + zl + ["-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ma_amd", "host"), src, "-o", READER_EXE,
                               "-L" + os.path.join(ROOT, "ma_amd"), "-lma_amd", "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"),
                               "-lpthread"] + (["-lz"] if zl else []))
    return READER_EXE


@pytest.mark.parametrize("name", ["reader_multi.fa", "reader_multi.fq", "reader_empty.fa", "reader_notfasta.txt",
                                  "reader_plusname.fq"])
def test_fasta_fastq_reader_matches_reference(tmp_path, name):
    """Multi-line records, CRLF, lower case, IUPAC codes, trailing junk, descriptions, empty reads, non-FASTA input:
    same reads / same error text as the reference's FileReader (tests/golden/reader/*.ref from oracle/_ref)."""
    exe = build_reader_exe()
    out = str(tmp_path / "o.txt")
    # the error text quotes the path as given: use the one the golden was made with
    subprocess.check_call([exe, os.path.join("reader", name), out], cwd=G)
    assert open(out).read() == open(os.path.join(G, "reader", name + ".ref")).read()


@pytest.mark.skipif(not os.path.exists("/usr/include/zlib.h"), reason="zlib headers not installed")
@pytest.mark.parametrize("name", ["reader_multi.fa", "reader_multi.fq", "reader_empty.fa", "reader_notfasta.txt", "reader_plusname.fq"])
@pytest.mark.parametrize("batch,block", [(1, 1), (2, 7), (3, 64), (1000, 1 << 20)])
def test_batch_reader_cuts_the_same_records(tmp_path, name, batch, block):
    """BatchFileReader (ma_batch_nodes.h) cuts a batch of records out of the stream under its lock and builds the reads
    outside of it: same reads, same error text as the per-record FileReader, whatever the batch size and wherever the block
    boundaries of the stream fall."""
    exe = build_reader_exe()
    out = str(tmp_path / "out.txt")
    subprocess.check_call([exe, os.path.join("reader", name), out, "batch", str(batch), str(block)], cwd=G)
    want = open(os.path.join(G, "reader", name + ".ref")).read()
    got = open(out).read()
    if "ERROR" in want:  # reads of the batch that holds the broken record are lost with it; the error text is the same
        assert got.splitlines()[-1] == want.splitlines()[-1]
        assert want.startswith("".join(l + "\n" for l in got.splitlines()[:-1]))
    else:
        assert got == want


@pytest.mark.parametrize("name", ["reader_multi.fa", "reader_multi.fq", "reader_plusname.fq"])
def test_gzip_input_matches_reference(tmp_path, name):
    """GzFileStream (WITH_ZLIB build of the reference): the compressed file yields the same reads."""
    exe = build_reader_exe()
    out = str(tmp_path / "o.txt")
    subprocess.check_call([exe, os.path.join("reader", name + ".gz"), out], cwd=G)
    assert open(out).read() == open(os.path.join(G, "reader", name + ".gz.ref")).read()


def test_sam_with_fastq_names_and_qualities(tmp_path):
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    pipe = gunzip_to(os.path.join(G, "small_ref.default.pipe.gz"), str(tmp_path / "p.pipe"))
    out = str(tmp_path / "o.sam")
    subprocess.check_call([exe, case, pipe, out, "0", os.path.join(G, "reader", "small24.fq")])
    want = gzip.open(os.path.join(G, "reader", "small24.fq.sam.gz"), "rt").read()
    assert open(out).read() == want
