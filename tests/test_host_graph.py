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
            os.path.join(ROOT, "ma_amd", "host", "ma_sam.h"), os.path.join(ROOT, "include", "ma_amd.h"),
            os.path.join(ROOT, "ma_amd", "host", "ma_batch_nodes.h"), os.path.join(ROOT, "ma_amd", "host", "ma_flat_sam.h"),
            os.path.join(ROOT, "ma_amd", "host", "ma_engine.h")]
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


def _same_alns_and_mq(got, want, alns=True):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        if alns:
            assert g["alns"] == w["alns"], "read %d alignments" % i
        assert len(g["mq"]) == len(w["mq"]), "read %d" % i
        for a, b in zip(g["mq"], w["mq"]):
            assert a == b, "read %d mapq record" % i


@pytest.mark.gpu
@pytest.mark.parametrize("preset,name", [("default", "small_ref.default.pipe"), ("illumina", "small_ref.illumina.pipe")])
@pytest.mark.parametrize("threads", [8, 64])
def test_unchanged_graph_with_many_threads_funnels_into_device_batches(tmp_path, gpu_device, preset, name, threads):
    """N copies of the setUpCompGraph chain over one shared reader, driven by simultaneousGet with N threads
    (export.cpp:84-126): the per-read execute() calls of all threads go through the device as batches (DeviceBatcher) and
    every read still gets exactly the reference's records."""
    import json
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "graph.out")
    line = subprocess.check_output([exe, case, preset, out, "threads", str(threads)]).decode().strip().splitlines()[-1]
    info = json.loads(line)
    want = parse_pipe_dump(os.path.join(G, name + ".gz"))
    assert info["reads"] == len(want)
    assert info["device_batches"] < info["reads"], info  # reads did share device batches
    _same_alns_and_mq(parse_pipe_dump(out), want)


@pytest.mark.gpu
@pytest.mark.parametrize("preset,name", [("default", "small_ref.default.pipe"), ("illumina", "small_ref.illumina.pipe")])
@pytest.mark.parametrize("threads,batch", [(4, 37), (16, 64), (3, 100000)])
def test_graph_with_prefetching_reader_needs_no_funnel(tmp_path, gpu_device, preset, name, threads, batch):
    """the same per-read graph with ONLY the reader node wrapped (PrefetchReader pulls `batch` reads ahead,
    sends them through all stages on the GPU and hands the graph threads reads that carry their ticket): a handful of graph
    threads, no read goes through the per-read funnel (the executable fails if one does), every read gets exactly the
    reference's records; batches smaller than, and larger than, the read set; end of input with reads still being handed out."""
    import json
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "graph.out")
    line = subprocess.check_output([exe, case, preset, out, "prefetch", str(threads), str(batch)]).decode().strip().splitlines()[-1]
    info = json.loads(line)
    want = parse_pipe_dump(os.path.join(G, name + ".gz"))
    assert info["reads"] == len(want)
    assert info["device_batches"] == (len(want) + batch - 1) // batch, info
    _same_alns_and_mq(parse_pipe_dump(out), want)


@pytest.mark.gpu
@pytest.mark.parametrize("preset,name", [("default", "small_ref.default.pipe"), ("illumina", "small_ref.illumina.pipe")])
def test_modules_take_over_mid_chain_from_plain_containers(tmp_path, gpu_device, preset, name):
    """Drop-in one stage at a time: every module is fed a container that does NOT come from the preceding MI355X module
    (rebuilt field by field, as the reference's modules would hand it over); each stage then runs on its own through
    ma_batch_set_segments / _seeds / _hsets.  Same records as the reference."""
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "graph.out")
    subprocess.check_call([exe, case, preset, out, "mixed"])
    _same_alns_and_mq(parse_pipe_dump(out), parse_pipe_dump(os.path.join(G, name + ".gz")))


@pytest.mark.gpu
@pytest.mark.parametrize("preset,name", [("default", "small_ref.default.pipe"), ("illumina", "small_ref.illumina.pipe")])
def test_soc_queue_pops_like_the_reference(tmp_path, gpu_device, preset, name):
    """SoCPriorityQueue::pop across the boundary (soc.h:240-284): strip index, score, ambiguity and the seeds of every
    popped strip equal the reference's SOC records (golden vectors G5)."""
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "socs.out")
    subprocess.check_call([exe, case, preset, out, "socs"])
    got = parse_pipe_dump(out)
    want = parse_pipe_dump(os.path.join(G, name + ".gz"))
    assert len(got) == len(want)
    n = 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert g["socs"] == w["socs"], "read %d: SoC queue differs" % i
        n += len(w["socs"])
    assert n > len(want) // 2


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 2, 3])
def test_multi_device_aligner_on_virtual_shards(tmp_path, gpu_device, shards):
    """MultiDeviceAligner: the device batches of a read set rotate over the index replicas (two workers per
    replica, one shared batch counter), results at the input positions; here the replicas are virtual shards on GPU 0."""
    import json
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "multi.out")
    info = json.loads(subprocess.check_output([exe, case, "default", out, "multi", str(shards)]).decode().strip().splitlines()[-1])
    want = parse_pipe_dump(os.path.join(G, "small_ref.default.pipe.gz"))
    assert info["reads"] == len(want) and info["device_batches"] == (len(want) + 36) // 37 and info["shards_used"] == shards
    _same_alns_and_mq(parse_pipe_dump(out), want, alns=False)


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 3])
def test_multi_device_aligner_flat_with_persistent_engines(tmp_path, gpu_device, shards):
    """MultiDeviceAligner::executeFlat -- the throughput form (flat results, no Alignment containers)
    over several replicas.  The aligner's engines persist: the SECOND run creates none (the executable fails if the process'
    engine count changed), its flat batches are in input order and hold the reference's records."""
    import json
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "multiflat.out")
    info = json.loads(subprocess.check_output([exe, case, "default", out, "multiflat", str(shards)]).decode().strip().splitlines()[-1])
    want = parse_pipe_dump(os.path.join(G, "small_ref.default.pipe.gz"))
    assert info["reads"] == len(want) and info["device_batches"] == info["device_batches_first_run"] == (len(want) + 36) // 37
    # two workers (engines) per replica, but no more workers than device batches
    assert info["engines_after_first_run"] == info["engines_after_second_run"] == min(2 * shards, info["device_batches"])
    assert info["shards_used"] == shards
    _same_alns_and_mq(parse_pipe_dump(out), want, alns=False)


@pytest.mark.gpu
@pytest.mark.parametrize("replicas", [2, 3])
def test_unchanged_graphs_feed_all_replicas_of_the_index(tmp_path, gpu_device, replicas):
    """replicateIndex attaches further copies of the index (one per GPU of the node; here virtual shards
    on device 0) to the ONE FMIndex / Pack the graph of export.cpp:99-126 holds.  The prefetching reader and the batch graph
    nodes rotate their device batches over the replicas -- the graph wiring is untouched -- and the records / SAM bytes are the
    reference's."""
    import json
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    env = dict(os.environ, MA_TEST_REPLICAS=str(replicas))
    want = parse_pipe_dump(os.path.join(G, "small_ref.default.pipe.gz"))
    out = str(tmp_path / "graph.out")
    line = subprocess.check_output([exe, case, "default", out, "prefetch", "6", "37"], env=env).decode().strip().splitlines()[-1]
    info = json.loads(line)
    assert info["reads"] == len(want) and info["device_batches"] == (len(want) + 36) // 37, info
    _same_alns_and_mq(parse_pipe_dump(out), want)
    sam = str(tmp_path / "batch.sam")
    info = json.loads(subprocess.check_output([exe, case, "default", sam, "batchgraph", "4", "50", "0"], env=env).decode().strip().splitlines()[-1])
    assert info["reads"] == len(want)
    sam_want = gzip.open(os.path.join(G, "small_ref.default.opt0.sam.gz"), "rt").read()

    def without_quality(line):  # the batch graph reads FASTQ text (quality column "III..."), the goldens' reads have none
        f = line.split("\t")
        if len(f) > 10:
            f[10] = "*"
        return "\t".join(f)

    # the batch graph writes whole batches in the order they finish: same records, compared as sorted lines
    assert sorted(without_quality(l) for l in open(sam).read().splitlines()) == sorted(sam_want.splitlines())


def build_index_store_exe():
    exe = os.path.join(ROOT, "tests", "emul", "index_store_test")
    src = exe + ".cpp"
    deps = [src] + [os.path.join(ROOT, "ma_amd", "host", h) for h in ("ma_modules.h", "ms_graph.h", "ma_sam.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "ma_amd", "host"), src, "-o", exe, "-L" + os.path.join(ROOT, "ma_amd"),
                               "-lma_amd", "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"), "-lpthread"])
    return exe


def test_load_index_rejects_truncated_or_foreign_files(tmp_path):
    """loadIndex does not trust the files: a truncated .sa / .bwt / .pac or a suffix array written with
    another sampling interval is refused with the reference's messages before anything is copied or uploaded
    (vRestoreBWT / vRestoreSuffixArray, fMIndex.h:555-663).  No GPU needed: validation comes first."""
    exe = build_index_store_exe()
    good = {ext: gzip.open(os.path.join(G, "small_ref." + ext + ".gz"), "rb").read() for ext in ("bwt", "sa", "pac")}
    ann = "61000 3 0\n0 chr1 none\n0 30000 0\n0 chr2 none\n30000 22000 0\n0 chr3 none\n52000 9000 0\n"

    def check(files):
        prefix = str(tmp_path / "idx")
        for ext in ("bwt", "sa", "pac"):
            with open(prefix + "." + ext, "wb") as f:
                f.write(files[ext])
        with open(prefix + ".ann", "w") as f:
            f.write(ann)
        return subprocess.check_output([exe, "loadcheck", prefix]).decode().strip()

    assert "Unexpected bad after reading suffix array" in check(dict(good, sa=good["sa"][:-8]))
    assert "non matching expected size" in check(dict(good, sa=good["sa"] + b"\0" * 8))
    other = bytearray(good["sa"])
    other[40:44] = (64).to_bytes(4, "little")
    assert "sampling interval 64" in check(dict(good, sa=bytes(other)))
    assert "Unexpected fail after reading BWT" in check(dict(good, bwt=good["bwt"][:-64]))
    assert "unexpected size" in check(dict(good, pac=good["pac"][:100]))
    bad_primary = bytearray(good["sa"])
    bad_primary[0] ^= 1
    assert "different primary" in check(dict(good, sa=bytes(bad_primary)))
    # the untouched files pass validation and only then need a device
    res = check(good)
    assert res == "ok" or "hip" in res.lower() or "device" in res.lower(), res


@pytest.mark.gpu
def test_gpu_built_index_is_stored_in_the_reference_file_formats(tmp_path, gpu_device):
    """f1, file side: storeIndex writes .bwt/.sa/.pac byte-identical to the reference's files for the same genome
    (tests/golden/small_ref.*), .ann/.amb in its text format, and the reference's own loaders accept them
    (oracle/_ref/ref_dump pipeidx, when that build is present: same pipeline dump as with its own index)."""
    exe = build_index_store_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    prefix = str(tmp_path / "idx")
    subprocess.check_call([exe, case, prefix, "small genome"])
    import json
    assert json.load(open(str(tmp_path / "small genome.json"))) == {"name": "small genome", "prefix": "idx", "type": "MA Genome",
                                                                    "version": {"major": 1, "minor": 0}}
    for ext in ("bwt", "sa", "pac"):
        want = gzip.open(os.path.join(G, "small_ref." + ext + ".gz"), "rb").read()
        assert open(prefix + "." + ext, "rb").read() == want, ext
    assert open(prefix + ".ann").read() == ("61000 3 0\n0 chr1 none\n0 30000 0\n0 chr2 none\n30000 22000 0\n"
                                            "0 chr3 none\n52000 9000 0\n")
    assert open(prefix + ".amb").read() == "61000 3 0\n"
    ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if os.path.exists(ref_dump):
        out = str(tmp_path / "ref.pipe")
        subprocess.check_call([ref_dump, "pipeidx", prefix, case, "default", "1", out])
        assert open(out).read() == gzip.open(os.path.join(G, "small_ref.default.pipe.gz"), "rt").read()
        # forward length not a multiple of 4 (no zero byte before the .pac check byte), tiny contig in the middle
        from ma_testlib import rand_genome, sample_reads, write_case
        g = rand_genome(77, [70001, 129, 40003])
        case2 = str(tmp_path / "odd.case")
        write_case(case2, g, sample_reads(g[:1], 40, 150, 78, sub=0.02) + sample_reads(g[2:], 40, 150, 79, sub=0.02))
        subprocess.check_call([exe, case2, str(tmp_path / "odd")])
        subprocess.check_call([ref_dump, "pipeidx", str(tmp_path / "odd"), case2, "default", "1", str(tmp_path / "a.pipe")])
        subprocess.check_call([ref_dump, "pipe", case2, "default", "1", str(tmp_path / "b.pipe")])
        assert open(str(tmp_path / "a.pipe")).read() == open(str(tmp_path / "b.pipe")).read()


@pytest.mark.gpu
def test_index_from_genome_fasta_with_n_runs(tmp_path, gpu_device):
    """buildIndexFromFasta + storeIndex on a FASTA with runs of N: contig table and hole records (.ann / .amb) as the
    reference's Pack::vAppendFASTA writes them (the random bases that replace the Ns differ by design: the reference seeds
    rand() with the time), and the reference loads the files."""
    import numpy as np
    from ma_testlib import rand_genome
    exe = os.path.join(ROOT, "tests", "emul", "index_store_test")
    rng = np.random.default_rng(5)
    g = rand_genome(78, [5000, 3001])
    fa = str(tmp_path / "g.fa")
    with open(fa, "w") as f:
        for k, c in enumerate(g):
            s = "".join("ACGT"[int(b)] for b in c)
            if k == 0:
                s = s[:100] + "N" * 37 + s[137:2000] + "NNN" + s[2003:]
            else:
                s = "N" * 5 + s[5:-2] + "NN"
            f.write(">ctg%d some description\n" % k)
            for i in range(0, len(s), 61):
                f.write(s[i:i + 61] + "\n")
    prefix = str(tmp_path / "fx")
    subprocess.check_call([exe, fa, prefix, "fasta genome", "fasta"])
    assert open(prefix + ".amb").read() == "8001 2 4\n100 37 N\n2000 3 N\n5000 5 N\n7999 2 N\n"
    ann = open(prefix + ".ann").read().split("\n")
    assert ann[1:5] == ["0 ctg0 none", "0 5000 2", "0 ctg1 none", "5000 3001 2"]
    ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if os.path.exists(ref_dump):
        subprocess.check_call([ref_dump, "fastapack", fa, str(tmp_path / "rf")])
        assert open(str(tmp_path / "rf.amb")).read() == open(prefix + ".amb").read()
        rann = open(str(tmp_path / "rf.ann")).read().split("\n")
        assert [l.split()[:2] for l in rann[1::2] if l] == [l.split()[:2] for l in ann[1::2] if l]  # gi, name
        assert rann[2::2] == ann[2::2]  # offset, length, holes
        # the packs agree outside the holes
        a, b = open(prefix + ".pac", "rb").read(), open(str(tmp_path / "rf.pac"), "rb").read()
        assert len(a) == len(b)
        holes = set()
        for o, n in ((100, 37), (2000, 3), (5000, 5), (7999, 2)):
            holes.update(range(o, o + n))
        for p in range(8001):
            if p not in holes:
                assert (a[p >> 2] >> ((~p & 3) << 1)) & 3 == (b[p >> 2] >> ((~p & 3) << 1)) & 3, p


@pytest.mark.gpu
def test_example_fastq_to_sam_end_to_end(tmp_path, gpu_device):
    """examples/ma_align.cpp: genome FASTA -> index on the GPU, FASTQ reads -> BatchAligner -> FileWriter; the SAM records
    equal what the reference's FileReader + modules + FileWriter wrote for the same files (tests/golden/reader)."""
    from ma_testlib import read_case
    exe = os.path.join(ROOT, "examples", "ma_align")
    src = exe + ".cpp"
    deps = [src] + [os.path.join(ROOT, "ma_amd", "host", h) for h in ("ma_sam.h", "ma_modules.h", "ms_graph.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        zl = os.path.exists("/usr/include/zlib.h")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall"] + (["-DMA_WITH_ZLIB"] if zl else []) +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ma_amd", "host"), src, "-o", exe,
                               "-L" + os.path.join(ROOT, "ma_amd"), "-lma_amd", "-Wl,-rpath," + os.path.join(ROOT, "ma_amd"),
                               "-lpthread"] + (["-lz"] if zl else []))
    contigs, _, names = read_case(gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case")))
    fa = str(tmp_path / "genome.fa")
    with open(fa, "w") as f:
        for nm, c in zip(names, contigs):
            f.write(">%s\n%s\n" % (nm, "".join("ACGT"[int(b)] for b in c)))
    out = str(tmp_path / "out.sam")
    subprocess.check_call([exe, fa, os.path.join(G, "reader", "small24.fq"), out, "default"])
    got = open(out).read().split("\n")
    want = gzip.open(os.path.join(G, "reader", "small24.fq.sam.gz"), "rt").read().split("\n")
    assert got[0] == "@SQ\tSN:chr1\tLN:30000"  # file-name constructor: tabs (fileWriter.h:385-400)
    assert [l for l in got if not l.startswith("@")] == [l for l in want if not l.startswith("@")]


@pytest.mark.gpu
@pytest.mark.parametrize("preset,options", [("default", 0), ("default", 1), ("default", 2), ("default", 3), ("default", 4), ("illumina", 0)])
@pytest.mark.parametrize("threads,batch", [(1, 1000), (3, 23)])
def test_batch_graph_nodes_write_the_reference_sam(tmp_path, gpu_device, preset, options, threads, batch):
    """The throughput form as graph nodes (ma_batch_nodes.h): BatchFileReader -> BatchAlign -> BatchFileWriter under
    promiseMe / simultaneousGet.  The reads come from FASTQ text, the results stay flat (no Alignment containers) and the SAM
    text is formatted from the flat view: the bytes are the reference FileWriter's (goldens written by the compiled
    reference) except for the quality column, which the goldens' FASTA-like reads do not have."""
    exe = build_exe()
    case = gunzip_to(os.path.join(G, "small.case.gz"), str(tmp_path / "small.case"))
    out = str(tmp_path / "batch.sam")
    stats = subprocess.check_output([exe, case, preset, out, "batchgraph", str(threads), str(batch), str(options)]).decode()
    want = gzip.open(os.path.join(G, "small_ref.%s.opt%d.sam.gz" % (preset, options)), "rt").read().splitlines()
    got = open(out).read().splitlines()

    def without_quality(line):
        f = line.split("\t")
        if len(f) > 10:
            assert set(f[10]) <= {"I"} and len(f[10]) > 0
            f[10] = "*"
        return "\t".join(f)

    got = [without_quality(l) for l in got]
    if threads == 1:
        assert got == want
    else:  # the batches of different graph threads finish in any order; inside a batch the order is the input's
        assert sorted(got) == sorted(want)
    assert '"reads": 128' in stats
