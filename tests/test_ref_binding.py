"""The MI355X modules on the reference's REAL types (ma_amd/host/ma_ref_binding.h): oracle/_ref/ref_graph_test is compiled
against the reference's own headers, linked against the reference compiled from its own sources (libma_ref.so) and against
libma_amd.so, and builds the chain of libMA::setUpCompGraph (export.cpp:104-108) with the reference's own promiseMe /
Pledge / simultaneousGet / containers / FileWriter.  CPU tests: it compiles and links here, fails loudly without a GPU,
and -- with every stage left to the reference -- reproduces the goldens (so the harness itself is sound).  GPU tests: all
five stages as ma_amd:: modules, and each ONE of them alone between the reference's CPU modules, give the reference's
records and the reference's SAM bytes."""
import gzip
import json
import os
import subprocess

import pytest

from ma_testlib import ROOT, gunzip_to, parse_pipe_dump

G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_graph_test")
HAVE_REFERENCE = os.path.isdir("/root/reference/libs/ma")


def build_exe():
    """Here (reference tree present): (re)build by the committed recipe.  On the GPU box the prebuilt binary travels."""
    if HAVE_REFERENCE:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "_ref/ref_graph_test"])
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/ref_graph_test not present (needs /root/reference to build)")
    return EXE


def small_case(tmp_path):
    return gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))


def same_dump(got_path, golden_name):
    got = parse_pipe_dump(got_path)
    want = parse_pipe_dump(os.path.join(G, golden_name + ".gz"))
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        for key in w:
            assert g[key] == w[key], "read %d: %s differs" % (i, key)


def test_binding_compiles_and_links_against_the_reference_headers():
    """One translation unit holds the reference's fileWriter.h / binarySeeding.h / ... AND ma_ref_binding.h: the modules
    derive from the reference's libMS::Module<> and exchange its containers (nothing is re-declared)."""
    if not HAVE_REFERENCE:
        pytest.skip("the reference tree is not on this machine")
    exe = build_exe()
    assert os.path.getmtime(exe) >= os.path.getmtime(os.path.join(ROOT, "ma_amd", "host", "ma_ref_binding.h"))
    src = open(os.path.join(ROOT, "ma_amd", "host", "ma_ref_binding.h")).read()
    includes = [l.split('"')[1] for l in src.splitlines() if l.startswith('#include "')]
    ours = [i for i in includes if not (i.startswith("ma/") or i.startswith("ms/") or i.startswith("util/"))]
    assert sorted(ours) == ["ma_amd.h", "ma_engine.h"], "the binding may only include reference headers + the C ABI + the batcher"
    assert "ms_graph.h" not in src.replace("ma_modules.h / ms_graph.h", "") and '#include "ma_modules.h"' not in src


def test_binding_fails_loudly_without_a_gpu(tmp_path):
    exe = build_exe()
    try:
        import ma_amd
        n = ma_amd.device_count()
    except Exception:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present: the no-GPU error path cannot be exercised here")
    out = subprocess.check_output([exe, "nogpu", small_case(tmp_path)]).decode()
    assert "std::runtime_error" in out


def test_harness_with_only_reference_modules_reproduces_the_goldens(tmp_path):
    """The same program with NO stage replaced is the reference itself: its dump and its SAM equal the committed goldens
    (made by ref_dump, which calls the modules directly instead of through pledges)."""
    exe = build_exe()
    case = small_case(tmp_path)
    out = str(tmp_path / "none.pipe")
    subprocess.check_call([exe, "pipe", case, "default", "1", out, "none"])
    assert open(out).read() == gzip.open(os.path.join(G, "small_ref.default.pipe.gz"), "rt").read()
    sam = str(tmp_path / "none.sam")
    subprocess.check_call([exe, "sam", case, "default", "1", sam, "none", "1"])
    assert open(sam).read() == gzip.open(os.path.join(G, "small_ref.default.opt0.sam.gz"), "rt").read()


@pytest.mark.gpu
@pytest.mark.parametrize("preset,seed,name", [("default", 1, "small_ref.default.pipe"), ("illumina", 1, "small_ref.illumina.pipe"),
                                              ("default", 7, "small_ref.default.seed7.pipe"), ("default+mems", 1, "small_ref.mems.pipe")])
def test_all_five_gpu_modules_in_the_reference_graph(tmp_path, gpu_device, preset, seed, name):
    """(i) export.cpp:104-108 built from the reference's promiseMe with all five stages as ma_amd:: modules: every stage
    record -- segments, extracted seeds, SoC pop order (the reference's own pop() on the queue the device filled), harmonized
    sets, alignments, mapping qualities -- equals the reference's."""
    exe = build_exe()
    out = str(tmp_path / "all.pipe")
    subprocess.check_call([exe, "pipe", small_case(tmp_path), preset, str(seed), out, "all"])
    assert open(out).read() == gzip.open(os.path.join(G, name + ".gz"), "rt").read()


@pytest.mark.gpu
@pytest.mark.parametrize("stages", ["seeding", "soc", "harm", "dp", "mq", "seeding,soc", "harm,dp", "dp,mq", "seeding,dp", "soc,harm,dp,mq"])
@pytest.mark.parametrize("preset,name", [("default", "small_ref.default.pipe"), ("illumina", "small_ref.illumina.pipe")])
def test_each_gpu_module_alone_between_the_reference_cpu_modules(tmp_path, gpu_device, stages, preset, name):
    """(ii) one stage (or a few) on the GPU, the others the reference's CPU modules: containers of the reference go into the
    MI355X module (uploaded) and its containers into the reference's next module, in both directions at every seam."""
    exe = build_exe()
    out = str(tmp_path / "mixed.pipe")
    subprocess.check_call([exe, "pipe", small_case(tmp_path), preset, "1", out, stages])
    assert open(out).read() == gzip.open(os.path.join(G, name + ".gz"), "rt").read()


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["default", "illumina"])
@pytest.mark.parametrize("threads", [1, 8, 48])
def test_reference_filewriter_behind_the_gpu_modules_writes_the_reference_sam(tmp_path, gpu_device, preset, threads):
    """The doAlign shape (execution-context.h:291-406): a shared volatile source, N graph copies under the reference's
    simultaneousGet, the reference's own FileWriter as sink.  The reads of all graph threads funnel into device batches; the
    SAM records equal the reference's (same bytes; with several threads the order of the records is the threads')."""
    exe = build_exe()
    sam = str(tmp_path / "gpu.sam")
    stats = json.loads(subprocess.check_output([exe, "sam", small_case(tmp_path), preset, "1", sam, "all", str(threads)]).decode())
    want = gzip.open(os.path.join(G, "small_ref.%s.opt0.sam.gz" % preset), "rt").read()
    got = open(sam).read()
    if threads == 1:
        assert got == want
    else:
        assert sorted(got.splitlines()) == sorted(want.splitlines())
        assert stats["device_batches"] < stats["reads"], "the graph threads' reads must share device batches"
    assert stats["reads_in_batches"] == stats["reads"]


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["default", "illumina"])
@pytest.mark.parametrize("threads,batch", [(1, 64), (4, 37), (16, 100000)])
def test_prefetching_reader_in_the_reference_graph_writes_the_reference_sam(tmp_path, gpu_device, preset, threads, batch):
    """VERDICT r3 item 4 on the reference's REAL types: the graph of export.cpp:99-126 under the reference's own
    promiseMe / simultaneousGet with ONE node changed -- the volatile source is wrapped into ma_amd::PrefetchReader, which pulls
    `batch` reads ahead, aligns them on the GPU and hands the graph threads TicketedQuery objects.  No read goes through the
    per-read funnel (device_batches of BinarySeeding stays 0), a handful of graph threads, the reference's FileWriter writes
    the reference's SAM bytes."""
    exe = build_exe()
    sam = str(tmp_path / "ahead.sam")
    env = dict(os.environ, MA_PREFETCH_BATCH=str(batch))
    stats = json.loads(subprocess.check_output([exe, "sam", small_case(tmp_path), preset, "1", sam, "all", str(threads), "8"], env=env).decode())
    want = gzip.open(os.path.join(G, "small_ref.%s.opt0.sam.gz" % preset), "rt").read()
    got = open(sam).read()
    if threads == 1:
        assert got == want
    else:
        assert sorted(got.splitlines()) == sorted(want.splitlines())
    assert stats["device_batches"] == 0 and stats["reads_in_batches"] == 0, "no read may go through the per-read funnel"
    assert stats["prefetched_reads"] == stats["reads"]
    assert stats["prefetched_batches"] == (stats["reads"] + batch - 1) // batch


@pytest.mark.gpu
@pytest.mark.parametrize("preset,options", [("default", 0), ("illumina", 0), ("default", 1), ("default", 2), ("illumina", 4)])
@pytest.mark.parametrize("threads", [1, 12])
def test_buffered_file_writer_on_the_references_types_writes_the_reference_sam(tmp_path, gpu_device, preset, options, threads):
    """ma_amd::BufferedFileWriter, a subclass of the reference's FileWriter that formats every read with a
    thread-private libMA::FileWriter (the reference's own code) and takes the shared lock once per 64 KB instead of once per read
    (fileWriter.cpp:141-145): the SAM goldens' bytes, for the option sets the goldens cover (soft clip, =/X cigars, NGMLR tags)."""
    exe = build_exe()
    sam = str(tmp_path / "buffered.sam")
    env = dict(os.environ, MA_PREFETCH_BATCH="37")
    stats = json.loads(subprocess.check_output([exe, "sam", small_case(tmp_path), preset, "1", sam, "all", str(threads), str(8 | 32 | options)], env=env).decode())
    want = gzip.open(os.path.join(G, "small_ref.%s.opt%d.sam.gz" % (preset, options)), "rt").read()
    got = open(sam).read()
    if threads == 1:
        assert got == want
    else:
        assert sorted(got.splitlines()) == sorted(want.splitlines())
    assert stats["prefetched_reads"] == stats["reads"]


@pytest.mark.gpu
def test_prefetching_reader_over_index_replicas_in_the_reference_graph(tmp_path, gpu_device):
    """VERDICT r4 item 1(b) on the reference's REAL types: ma_amd::replicateIndex attaches two more copies of the index (virtual
    shards on device 0; one per GPU on a node) to the attached Pack / FMIndex pair, the prefetching reader rotates its device
    batches over the three, the reference's graph, modules' wiring and FileWriter are untouched: the reference's SAM records."""
    exe = build_exe()
    sam = str(tmp_path / "ahead.sam")
    env = dict(os.environ, MA_PREFETCH_BATCH="23", MA_TEST_REPLICAS="3")
    stats = json.loads(subprocess.check_output([exe, "sam", small_case(tmp_path), "default", "1", sam, "all", "6", "8"], env=env).decode())
    want = gzip.open(os.path.join(G, "small_ref.default.opt0.sam.gz"), "rt").read()
    assert sorted(open(sam).read().splitlines()) == sorted(want.splitlines())
    assert stats["device_batches"] == 0 and stats["prefetched_reads"] == stats["reads"]
    assert stats["prefetched_batches"] == (stats["reads"] + 22) // 23


@pytest.mark.gpu
@pytest.mark.parametrize("threads,batch", [(1, 50), (8, 100000)])
def test_prefetching_reader_around_the_references_own_filereader(tmp_path, gpu_device, threads, batch):
    """The line INTEGRATION.md adds at export.cpp:83, literally: ma_amd::PrefetchReader<FileStream> around the reference's OWN
    FileReader (fileReader.cpp:37-203) over a FASTA file, its own FileWriter as sink: the goldens' SAM bytes."""
    from ma_testlib import read_case
    exe = build_exe()
    case = small_case(tmp_path)
    _, reads, _ = read_case(case)
    fa = str(tmp_path / "reads.fa")
    with open(fa, "w") as f:
        for i, r in enumerate(reads):
            f.write(">r%d\n%s\n" % (i, "".join("ACGTN"[min(int(b), 4)] for b in r)))
    sam = str(tmp_path / "file.sam")
    env = dict(os.environ, MA_PREFETCH_BATCH=str(batch))
    stats = json.loads(subprocess.check_output([exe, "sam", case, "default", "1", sam, "all", str(threads), "16", fa], env=env).decode())
    want = gzip.open(os.path.join(G, "small_ref.default.opt0.sam.gz"), "rt").read()
    got = open(sam).read()
    if threads == 1:
        assert got == want
    else:
        assert sorted(got.splitlines()) == sorted(want.splitlines())
    assert stats["device_batches"] == 0 and stats["prefetched_reads"] == len(reads)


@pytest.mark.gpu
def test_mixed_graph_writes_the_reference_sam(tmp_path, gpu_device):
    exe = build_exe()
    for stages in ("dp", "seeding,soc", "harm,dp,mq"):
        sam = str(tmp_path / "mixed.sam")
        subprocess.check_call([exe, "sam", small_case(tmp_path), "default", "1", sam, stages, "1", "3"])
        assert open(sam).read() == gzip.open(os.path.join(G, "small_ref.default.opt3.sam.gz"), "rt").read()
