"""SURVEY 8(f) row f4 in the host layer: SmallInversions, PairedReads, PairedFileWriter, PairedFileReader
(ma_amd/host/ma_modules.h, ma_sam.h) against what the compiled reference produced (tests/golden/f4.*, reader/mates.*)."""
import gzip
import os
import subprocess

import pytest

from ma_testlib import ROOT, gunzip_to

G = os.path.join(ROOT, "tests", "golden")
# (preset, search inversions, paired, Z Drop Inversions, SAM options) -- make_golden.py F4_CONFIGS
F4_CONFIGS = [("default", 1, 0, 100, 0), ("default", 1, 1, 100, 0), ("illumina", 0, 1, 100, 3), ("default", 1, 1, 40, 1)]


def build(name):
    exe = os.path.join(ROOT, "tests", "emul", name)
    src = exe + ".cpp"
    deps = [src, os.path.join(ROOT, "include", "ma_amd.h")] + [os.path.join(ROOT, "ma_amd", "host", h)
                                                              for h in ("ma_sam.h", "ma_modules.h", "ms_graph.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        zl = ["-DMA_WITH_ZLIB"] if os.path.exists("/usr/include/zlib.h") else []  # same flags as tests/test_sam_writer.py
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall"] + zl + ["-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ma_amd", "host"), src, "-o", exe,
                               "-L" + os.path.join(ROOT, "ma_amd"), "-lma_amd", "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"),
                               "-lpthread"] + (["-lz"] if zl else []))
    return exe


def same_text(got_path, want_gz, what):
    got = open(got_path).read().split("\n")
    want = gzip.open(want_gz, "rt").read().split("\n")
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, "%s line %d differs" % (what, i)
    assert len(got) == len(want), what


@pytest.mark.parametrize("cfg", F4_CONFIGS)
def test_pairing_and_sam_writers_match_reference(tmp_path, cfg):
    """PairedReads + PairedFileWriter (and FileWriter with inversion records) on the reference's own per-mate lists."""
    preset, inv, paired, zd, opt = cfg
    nm = "f4.%s.inv%d.pair%d.zd%d.opt%d" % cfg
    exe = build("f4_test")
    case = gunzip_to(os.path.join(G, "f4.case.gz"), str(tmp_path / "f4.case"))
    dump = gunzip_to(os.path.join(G, nm + ".f4.gz"), str(tmp_path / "ref.f4"))
    subprocess.check_call([exe, case, dump, preset, str(paired), str(opt), str(tmp_path / "o.f4"), str(tmp_path / "o.sam")])
    same_text(str(tmp_path / "o.f4"), os.path.join(G, nm + ".f4.gz"), "f4 dump")
    same_text(str(tmp_path / "o.sam"), os.path.join(G, nm + ".sam.gz"), "SAM")


@pytest.mark.parametrize("rc", [0, 1])
def test_paired_file_reader_matches_reference(tmp_path, rc):
    """Two FASTQ streams -> mate pairs; the second mate reverse-complemented (with its qualities) when
    "Paired Mate - Mate Pair" is set; the shorter file ends the pairs."""
    exe = build("reader_test")
    out = str(tmp_path / "o.txt")
    subprocess.check_call([exe, os.path.join("reader", "mates_1.fq"), out, os.path.join("reader", "mates_2.fq"), str(rc)], cwd=G)
    assert open(out).read() == open(os.path.join(G, "reader", "mates.rc%d.ref" % rc)).read()


def test_presets_of_the_parameter_set_manager(tmp_path):
    src = tmp_path / "p.cpp"
    src.write_text('''#include "ma_modules.h"
#include <cstdio>
int main(){ libMA::ParameterSetManager m; const char* n[]={"default","illumina","illuminapaired","pacbio","nanopore"};
for(auto k:n){ m.setSelected(k); auto p=m.getSelected(); printf("%s %d %d %d %d %d %d\\n",k,p->seeding_technique,p->max_ambiguity,
p->min_num_soc,p->max_num_soc,p->max_supplementary,p->use_paired_reads);}
try{ m.setSelected("nope"); }catch(const std::runtime_error& e){ printf("%s\\n", e.what()); } return 0; }''')
    exe = str(tmp_path / "p")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ma_amd", "host"),
                           str(src), "-o", exe, "-L" + os.path.join(ROOT, "ma_amd"), "-lma_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"), "-lpthread"])
    out = subprocess.check_output([exe]).decode().split("\n")
    # parameter.h:1081-1104
    assert out[:5] == ["default 0 100 1 30 1 0", "illumina 1 500 10 20 1 0", "illuminapaired 1 500 10 20 1 1",
                       "pacbio 0 100 5 30 100 0", "nanopore 1 100 5 30 100 0"]
    assert "can not be found" in out[5]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", F4_CONFIGS)
def test_graph_with_small_inversions_and_paired_reads_matches_reference(tmp_path, gpu_device, cfg):
    """setUpCompGraph / setUpCompGraphPaired (export.cpp:72-202) of MI355X modules: SmallInversions' DP on the GPU
    (ma_ksw_batch + ma_pack_extract), PairedReads, (Paired)FileWriter; lists and SAM identical to the reference's."""
    preset, inv, paired, zd, opt = cfg
    nm = "f4.%s.inv%d.pair%d.zd%d.opt%d" % cfg
    exe = build("f4_graph_test")
    case = gunzip_to(os.path.join(G, "f4.case.gz"), str(tmp_path / "f4.case"))
    subprocess.check_call([exe, case, preset, "1", str(tmp_path / "o.f4"), str(inv), str(paired), str(zd), str(tmp_path / "o.sam"),
                           str(opt)])
    same_text(str(tmp_path / "o.f4"), os.path.join(G, nm + ".f4.gz"), "f4 dump")
    same_text(str(tmp_path / "o.sam"), os.path.join(G, nm + ".sam.gz"), "SAM")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", F4_CONFIGS)
def test_batch_aligner_with_small_inversions_and_paired_reads(tmp_path, gpu_device, cfg):
    """The throughput API: BatchAligner::execute / executePaired put all reads through one device batch and all
    inversion DP through one ma_ksw_batch launch; same lists and SAM text as the reference's per-read graph."""
    preset, inv, paired, zd, opt = cfg
    nm = "f4.%s.inv%d.pair%d.zd%d.opt%d" % cfg
    exe = build("f4_graph_test")
    case = gunzip_to(os.path.join(G, "f4.case.gz"), str(tmp_path / "f4.case"))
    subprocess.check_call([exe, case, preset, "1", str(tmp_path / "o.f4"), str(inv), str(paired), str(zd), str(tmp_path / "o.sam"),
                           str(opt), "batch"])
    same_text(str(tmp_path / "o.sam"), os.path.join(G, nm + ".sam.gz"), "SAM")
    want = gzip.open(os.path.join(G, nm + ".f4.gz"), "rt").read().split("\n")
    if paired:  # the batch dump holds the pair records only
        want = [l for l in want if not l.startswith(("f ", "FIN"))]
    got = open(str(tmp_path / "o.f4")).read().split("\n")
    assert got == want
