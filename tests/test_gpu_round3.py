"""Round-3 GPU tests: the stage inputs added for the binding on the reference's real types (SoC queue state in / out,
MappingQuality alone), and the regressions of ADVICE round 2 (task-array overflow of the area-parallel seeding kernel, SMEM
lists of odd read lengths, a second ma_dp_batch on the same batch)."""
import numpy as np
import pytest

from ma_testlib import rand_genome, sample_reads

pytestmark = pytest.mark.gpu


def _batch(idx, reads, technique=0, preset="default"):
    import ma_amd
    P = ma_amd.Params.preset(preset)
    if technique is not None:
        P.seeding_technique = technique
    b = ma_amd.Batch(idx, P, len(reads), sum(len(r) for r in reads) + 64)
    b.set_reads(reads)
    return b


def _all_records(b):
    return [b.segments(), b.seeds(), b.hsets(), b.alignments(), b.mapq_alignments()]


def _same(got, want):
    for gs, ws in zip(got, want):
        for x, y in zip(gs, ws):
            assert np.array_equal(x, y)


@pytest.fixture(scope="module")
def genome_and_index(gpu_device):
    import ma_amd
    g = rand_genome(41, [700000, 260000], repeat_unit=300, repeat_copies=60, repeat_div=0.08)
    idx = ma_amd.Index.build(g)
    yield g, idx
    idx.close()


def test_soc_queue_state_round_trip(genome_and_index):
    """ma_batch_get_soc_heap -> ma_batch_set_soc_heap -> ma_chain_batch (harmonization only) gives the harmonized sets of the
    fused path; popping the heap array with libstdc++'s pop_heap rule (here: the device's own pop order) agrees with
    ma_batch_get_socs."""
    g, idx = genome_and_index
    reads = sample_reads(g, 300, 150, 5, sub=0.02) + sample_reads(g, 12, 3000, 6, sub=0.01, ins=0.005, dele=0.005)
    b = _batch(idx, reads)
    b.align()
    b.sync()
    want = b.hsets()
    want_alns = b.mapq_alignments()
    pop_off, pop, _, _ = b.socs(heap=False)
    heap_off, heap, seed_off, sorted_seeds = b.socs(heap=True)
    assert np.array_equal(pop_off, heap_off)
    for r in range(len(reads)):
        a, e = int(pop_off[r]), int(pop_off[r + 1])
        if e > a:
            # the same strips, and the first pop is the heap's root
            assert sorted(map(tuple, pop[a:e].tolist())) == sorted(map(tuple, heap[a:e].tolist()))
            assert tuple(pop[a].tolist()) == tuple(heap[a].tolist())
    b.close()
    b2 = _batch(idx, reads)
    b2.set_soc_heap(heap_off, heap, seed_off, sorted_seeds)
    b2.chain()
    b2.sync()
    got = b2.hsets()
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    b2.dp()
    b2.sync()
    for x, y in zip(b2.mapq_alignments(), want_alns):
        assert np.array_equal(x, y)
    b2.close()


def test_mapping_quality_alone_on_uploaded_alignments(genome_and_index):
    """ma_batch_set_alignments: the alignments of a NeedlemanWunsch that ran elsewhere, in its output order -> the
    MappingQuality kernel alone -> the selection, flags and qualities of the fused path."""
    g, idx = genome_and_index
    reads = (sample_reads(g, 400, 150, 7, sub=0.02) + sample_reads(g, 10, 2500, 8, sub=0.01, ins=0.005, dele=0.005)
             + sample_reads(g, 40, 150, 9, random_frac=1.0))
    for technique, preset in ((0, "default"), (1, "illumina")):
        b = _batch(idx, reads, technique=None, preset=preset)
        b.align()
        b.sync()
        off, alns, ops = b.alignments()
        want = b.mapq_alignments()
        aligned = b.counts()["aligned_reads"]
        b.close()
        b2 = _batch(idx, reads, technique=None, preset=preset)
        b2.set_alignments(off, alns, ops[: 2 * int(alns["n_ops"].sum())] if len(alns) else ops[:0])
        b2.sync()
        got = b2.mapq_alignments()
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
        again = b2.alignments()
        assert np.array_equal(again[0], off) and np.array_equal(again[1][["begin_ref", "end_ref", "score", "n_ops"]],
                                                                  alns[["begin_ref", "end_ref", "score", "n_ops"]])
        assert b2.counts()["aligned_reads"] == aligned
        b2.close()


def test_second_dp_stage_on_the_same_batch_does_not_double_the_counters(genome_and_index):
    """the DP stage owns its counters; running it twice must leave the same counts, sizes and downloads."""
    g, idx = genome_and_index
    reads = sample_reads(g, 500, 150, 11, sub=0.02) + sample_reads(g, 6, 4000, 12, sub=0.01, ins=0.005, dele=0.005)
    b = _batch(idx, reads)
    b.align()
    b.sync()
    c0, r0, k0 = b.counts(), _all_records(b), b.counters()
    b.dp()
    b.sync()
    c1, r1, k1 = b.counts(), _all_records(b), b.counters()
    assert c0 == c1
    assert np.array_equal(k0[4:], k1[4:])
    _same(r1, r0)
    b.close()


def test_task_array_overflow_falls_back_to_the_read_per_lane_kernel(genome_and_index, monkeypatch):
    """reads made of N only expand the area tree down to single bases, a level then holds more areas than
    the task array; the levels queued behind the overflow must not touch the unwritten slots and the batch must come out as
    the read-per-lane kernel computes it."""
    g, idx = genome_and_index
    rng = np.random.default_rng(3)
    reads = [np.full(6000, 4, dtype=np.uint8) for _ in range(3)]
    mostly_n = np.full(9000, 4, dtype=np.uint8)
    mostly_n[rng.integers(0, 9000, 300)] = rng.integers(0, 4, 300)
    reads.append(mostly_n)
    reads += sample_reads(g, 4, 5000, 13, sub=0.01)
    monkeypatch.setenv("MA_SEED_TASKS", "0")
    b = _batch(idx, reads)
    b.align()
    b.sync()
    want = _all_records(b)
    b.close()
    monkeypatch.setenv("MA_SEED_TASKS", "1")
    b = _batch(idx, reads)
    b.align()
    b.sync()
    _same(_all_records(b), want)
    b.close()


@pytest.mark.parametrize("length", [151, 149, 37])
def test_smem_lists_of_odd_read_lengths(genome_and_index, monkeypatch, length):
    """the 16-byte SMEM list entries of a lane start at lane * stride; the stride must stay a multiple of 16 bytes
    for odd read lengths too (151 bp Illumina reads).  Packed and 40-byte entries give the same records."""
    g, idx = genome_and_index
    reads = sample_reads(g, 700, length, 14, sub=0.02) + sample_reads(g, 50, length, 15, sub=0.05, n_rate=0.02)
    monkeypatch.setenv("MA_SMEM_COMPACT", "0")
    b = _batch(idx, reads, technique=1)
    b.align()
    b.sync()
    want = _all_records(b)
    b.close()
    monkeypatch.delenv("MA_SMEM_COMPACT")
    b = _batch(idx, reads, technique=1)
    b.align()
    b.sync()
    _same(_all_records(b), want)
    b.close()


def test_bench_four_ranks_long_reads_both_scaling_modes(gpu_device):
    """the multi-rank flow of bench.py with LONG reads -- where every rank's device batch holds GBs of
    pools and the DP stage runs its kernel classes on side streams -- as four ranks on one device (gloo; the driver's runs
    use one GPU per rank over RCCL), in both scaling modes.  weak: every rank aligns its own reads, the job aligns four
    times the reads of one rank; strong: ONE read set cut into four contiguous blocks gives exactly the aligned reads of
    the single process."""
    from test_gpu_round2 import _bench
    common = ["--workload", "10kb", "--genome-scale", "0.02", "--steps", "2", "--warmup", "1", "--reads-per-step", "2000",
              "--cpu-sample", "0", "--boundary-reads", "0", "--overlap", "0"]
    one = _bench(common + ["--gpus", "1"])
    weak = _bench(common + ["--gpus", "4"], nproc=4, env={"MA_BENCH_ONE_DEVICE": "1"})
    strong = _bench(common + ["--gpus", "4", "--scaling", "strong"], nproc=4, env={"MA_BENCH_ONE_DEVICE": "1"})
    w1, ww, ws = one["config"]["workloads"][0], weak["config"]["workloads"][0], strong["config"]["workloads"][0]
    assert weak["n_gpus"] == 4 and strong["n_gpus"] == 4 and weak["scaling"] == "weak" and strong["scaling"] == "strong"
    assert w1["aligned_reads"] > 0.95 * 4000
    assert ws["aligned_reads"] == w1["aligned_reads"]
    assert 3.8 * w1["aligned_reads"] < ww["aligned_reads"] < 4.2 * w1["aligned_reads"]
    # value is the whole job's rate: reads of all ranks over the slowest rank's time
    assert ww["value"] > 0 and ws["value"] > 0


@pytest.mark.parametrize("env", [{"MA_CHAIN_WAVE_SORT": "0"}, {"MA_WSORT_MIN": "20", "MA_WSORT_SMALL": "80"}, {"MA_WSORT_MIN": "100"},
                                 {"MA_WSORT_MIN": "20", "MA_WSORT_SMALL": "100"}, {"MA_WSORT_MIN": "20", "MA_SOC_WAVE": "0"},
                                 {"MA_WSORT_MIN": "20", "MA_SOC_WAVE": "2"}, {"MA_WSORT_MIN": "20"},
                                 {"MA_DP_ONE_STREAM": "1"}, {"MA_KSW_SCRATCH_MB": "64"}, {"MA_STITCH_WAVE": "0"}])
def test_long_read_stage_variants_give_identical_results(gpu_device, monkeypatch, env):
    """Round-3 variants of the long-read stages forced through their hooks on one read set with repeats (many equal deltas
    and reference positions = ties in the sweep's sorts): the sweep's sorts inside the lane kernels / as wave-cooperative
    kernels with both launch sizes -- and (round 6) the launch whose arrays stay in global memory -- exercised (thresholds moved down to
    test-sized reads), the window sweep by one wavefront per read (round 6; with MA_WSORT_MIN=20 on every read of more than 20 seeds), by
    lanes only (MA_SOC_WAVE=0) and through the wave kernel's fallback (2), the DP classes on one stream / on
    their own streams, a tiny DP scratch budget (every class split into tiers), the walk of the long alignments by one lane
    each instead of one wavefront each.  Every stage record equals the default's."""
    import ma_amd
    g = rand_genome(53, [900000, 400000], repeat_unit=250, repeat_copies=120, repeat_div=0.04)
    reads = (sample_reads(g, 40, 6000, 91, sub=0.01, ins=0.005, dele=0.005) + sample_reads(g, 6, 20000, 92, sub=0.03, ins=0.03, dele=0.04)
             + sample_reads(g, 200, 150, 93, sub=0.01) + sample_reads(g, 10, 2500, 94, sub=0.05, ins=0.02, dele=0.02))
    idx = ma_amd.Index.build(g)

    def run():
        b = _batch(idx, reads)
        b.align()
        b.sync()
        out = _all_records(b)
        n_seeds = np.diff(b.seeds()[0].astype(np.int64))
        b.close()
        return out, n_seeds

    want, n_seeds = run()
    assert (n_seeds > 100).sum() >= 3 and (n_seeds > 60).sum() >= 10 and ((n_seeds > 20) & (n_seeds <= 60)).sum() >= 3, \
        "the read set must reach the sort thresholds of the variants"
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got, _ = run()
    _same(got, want)
    idx.close()


@pytest.mark.parametrize("min_amb", [0, 2])
def test_smem_twin_entries_are_dropped_without_changing_the_segments(gpu_device, monkeypatch, min_amb):
    """The SMEM backward phase drops list entries whose interval equals that of the entry before them (seeding.h,
    seed_apply); every stage record equals that of the lists kept entry by entry (MA_SMEM_MERGE=0) -- on unique reads,
    reads out of repeats (many distinct intervals survive for long), reads with Ns and random reads -- and the extension
    counter shows the saving.  uiMinAmbiguity > 0 keeps the full lists (a failed twin may be pushed there)."""
    import ma_amd
    g = rand_genome(67, [600000, 300000], repeat_unit=180, repeat_copies=150, repeat_div=0.03)
    reads = (sample_reads(g, 1500, 150, 21, sub=0.02) + sample_reads(g, 300, 151, 22, sub=0.05, n_rate=0.02)
             + sample_reads(g, 200, 100, 23, random_frac=1.0) + sample_reads(g, 40, 1200, 24, sub=0.01, ins=0.005, dele=0.005))
    idx = ma_amd.Index.build(g)

    def run():
        P = ma_amd.Params.preset("illumina")
        P.seeding_technique = 1
        P.min_ambiguity = min_amb
        b = ma_amd.Batch(idx, P, len(reads), sum(len(r) for r in reads) + 64)
        b.set_reads(reads)
        b.align()
        b.sync()
        out = _all_records(b)
        steps = int(b.counters()[0])  # extend_backward steps
        b.close()
        return out, steps

    monkeypatch.setenv("MA_SMEM_MERGE", "0")
    want, steps_full = run()
    monkeypatch.delenv("MA_SMEM_MERGE")
    got, steps = run()
    _same(got, want)
    assert len(want[0][1]) > 2000
    if min_amb == 0:
        assert steps < 0.6 * steps_full
    else:
        assert steps == steps_full
    idx.close()


def test_long_read_seeding_kmer_jump_gives_the_same_segments(gpu_device, monkeypatch):
    """The read-per-lane kernel for reads in HBM (k_seed_long) takes the first K-1 steps of a run from the K-mer table, the
    key assembled out of two 16-byte blocks of the reads array (seeding.h, seed_jump): same records as the walk step by
    step -- with Ns inside K-mers, centres within K bases of the read ends, and reads at both ends of the reads array (where
    the blocks are taken flush with the array instead of aligned)."""
    import ma_amd
    g = rand_genome(71, [800000, 300000], repeat_unit=200, repeat_copies=80, repeat_div=0.05)
    reads = (sample_reads(g, 3, 40, 31, sub=0.0) + sample_reads(g, 60, 3000, 32, sub=0.01, ins=0.003, dele=0.003)
             + sample_reads(g, 30, 700, 33, sub=0.03, n_rate=0.01) + sample_reads(g, 200, 150, 34, sub=0.02) + sample_reads(g, 3, 33, 35, sub=0.0))
    idx = ma_amd.Index.build(g)
    monkeypatch.setenv("MA_SEED_TASKS", "0")

    def run():
        b = _batch(idx, reads)
        b.align()
        b.sync()
        out = _all_records(b)
        steps = int(b.counters()[0])
        b.close()
        return out, steps

    monkeypatch.setenv("MA_SEED_LONG_JUMP", "0")
    want, steps_walk = run()
    monkeypatch.delenv("MA_SEED_LONG_JUMP")
    got, steps = run()
    _same(got, want)
    assert steps < 0.9 * steps_walk
    # 256 resident lanes for 296 reads: the lanes fetch a second read from the queue when their first one is done
    monkeypatch.setenv("MA_SEED_LANES", "256")
    got, steps2 = run()
    _same(got, want)
    assert steps2 == steps
    idx.close()
