#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the MI355X seed-and-extend path on BASELINE.json's workloads.

A "step" is one pass of the whole hot path (FMD seeding -> seed extraction -> SoC/harmonization ->
banded-DP gap fill/extension -> mapping quality) over one batch of synthetic reads that is already
resident in HBM.  The metric is quoted on "150bp & 10kb" reads, so the default invocation runs, against ONE
GRCh38-sized synthetic genome (24 contigs, 3.09 Gnt, planted repeat families) and in one process per GPU:

  C2  150 bp Illumina-like reads (0.5 % substitutions), 1 M reads per step            -> `value` (headline, history)
  C3  10 kb CCS-like reads (0.4 / 0.3 / 0.3 % sub / ins / del), 5 steps of 200 k reads = the 1 M reads of configs[2]
  C5  50 kb ONT-like reads (3 / 3 / 4 %), 4 steps of 20 k reads (stress shape)

plus C2 once more under the Illumina preset (SMEM seeding) and, at N = 1, BASELINE.md section 3's sanity anchor
on C1 (oracle vs the compiled reference vs the GPU path on `ecoli_like`, 1 k reads).  Every workload runs three legs, each
with its own timed region (barrier + synchronize on both sides, MAX over ranks):

  single stream      one batch at a time, reads and results resident in HBM: undisturbed launches -> the roofline block,
                     the CPU baseline (the compiled reference on the host cores, N = 1 only) and the parity check
  device resident    several batches in flight (own streams + host threads), reads and results resident in HBM
  host to host       BASELINE.md section 3's region: every step's reads start in page-locked HOST memory, the flat result arrays
                     (offsets, alignment headers, ops) end in page-locked HOST memory, several batches in flight so that
                     the copies of one batch hide behind the kernels of another

The line's top-level value / ms_per_step / roofline are those of the HOST-TO-HOST leg of C2 (config.value_is says so);
value_150bp_device_resident, value_10kb, value_50kb, value_150bp_illumina ... are top-level scalars of the same line; the
full per-workload blocks (and the boundary leg) go to a side file (config.detail_file) and to stderr, so that the printed
line stays short.  --workload {150bp,10kb,50kb,illumina} or an explicit --read-len runs a single workload.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL, used only for the barrier and the
max-over-ranks time); the index is replicated, there is no data-path collective.  --scaling weak (default): every rank
aligns its own steps x reads_per_step reads; --scaling strong: ONE read set of steps x reads_per_step reads is
partitioned into contiguous blocks over the ranks.  Prints ONE JSON line on rank 0.  `--gpus N` without a torchrun
environment starts the N ranks itself (python -m torch.distributed.run ... bench.py ...) before anything touches a GPU
and exits with their status; inside a torchrun environment WORLD_SIZE must equal --gpus.
"""
import argparse
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616,
          64444167, 46709983, 50818468, 156040895, 57227415]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, a wave64 VALU instruction occupies a SIMD for 2 cycles (full-rate ops) at 2.4 GHz
CHIP_VALU_PEAK_GINST = 256 * 4 * 2.4 / 2
STAGES = ["k_seed", "k_seed_rows+k_lf_walk+k_seed_final", "k_chain", "k_dp_enum", "k_ksw", "k_stitch+k_finish"]

WORKLOADS = {
    # name: read_len, sub, ins, dele, reads seed, reads/step, default steps, warm-up, CPU sample (reads)
    "150bp": dict(read_len=150, sub=0.005, ins=0.0, dele=0.0, seed=11, reads_per_step=1000000, steps=None, warmup=None,
                  cpu_sample=None, baseline_config="configs[1] (C2)"),
    # (h2h_inflight 1: one batch object, its next upload and last download beside its own kernels -- 545 k reads/s against 459 k with two
    # batches in flight, whose seeding kernels crawl beside each other's persistent DP waves: profiles/r06_h2h_inflight.txt)
    "10kb": dict(read_len=10000, sub=0.004, ins=0.003, dele=0.003, seed=12, reads_per_step=200000, steps=5, warmup=1,
                 cpu_sample=15360, baseline_config="configs[2] (C3)", h2h_inflight=1),
    "50kb": dict(read_len=50000, sub=0.03, ins=0.03, dele=0.04, seed=13, reads_per_step=20000, steps=4, warmup=1,
                 cpu_sample=2048, baseline_config="configs[4] (C5 shape, one GPU)"),
    # C2's reads under the Illumina preset (SMEM seeding, parameter.h:1083-1087): SURVEY 8(d) asks for this row beside every
    # 150 bp table.
    "illumina": dict(read_len=150, sub=0.005, ins=0.0, dele=0.0, seed=11, reads_per_step=1000000, steps=10, warmup=1,
                     cpu_sample=20000, baseline_config="configs[1] (C2), Illumina preset", preset="illumina"),
    # configs[2] / configs[4] name "PacBio-CCS-like" and "ONT-like" reads: the same reads under the reference's PacBio and
    # Nanopore parameter sets (parameter.h:1096-1104: >= 5 strips per read, up to 100 supplementary alignments; Nanopore: SMEM
    # seeding).  One batch at a time, half a step's reads, a parity sample against the oracle; the compiled reference is timed on
    # that sample under the same parameter set (cpu_baseline.kind "reference").
    "10kb_pacbio": dict(read_len=10000, sub=0.004, ins=0.003, dele=0.003, seed=12, reads_per_step=100000, steps=2, warmup=1,
                        cpu_sample=2048, baseline_config="configs[2] (C3), PacBio preset", preset="pacbio", single_only=True),
    "50kb_nanopore": dict(read_len=50000, sub=0.03, ins=0.03, dele=0.04, seed=13, reads_per_step=10000, steps=2, warmup=1,
                          cpu_sample=512, baseline_config="configs[4] (C5 shape, one GPU), Nanopore preset", preset="nanopore", single_only=True),
}


def kernel_source_hash():
    """Identifies the build of the kernels: sha256 over ma_amd/csrc/*.{h,hip} (sorted by name), first 16 hex digits.  The
    PMC passes under profiles/ carry the hash of the sources they were collected with (tools/pmc_summarize.py)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ma_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_replay(read_len, reads_per_step, preset, genome_scale, stage):
    """HBM bytes and VALU instructions per launch of `stage` from the committed rocprofv3 PMC passes of the same workload
    (they are not measured in this process).  Refused -- null + reason -- when no pass of this workload is committed or when
    the pass was collected with other kernel sources than the ones libma_amd.so is built from now."""
    cur = kernel_source_hash()
    reason = "no PMC pass of this workload under profiles/"
    for src in (os.path.join("profiles", "r06_pmc_traffic.json"), os.path.join("profiles", "r05_pmc_traffic.json"),
                os.path.join("profiles", "r04_pmc_traffic.json"), os.path.join("profiles", "r03_pmc_traffic.json"),
                os.path.join("profiles", "r02_pmc_traffic.json")):
        try:
            with open(os.path.join(ROOT, src)) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            continue
        for pt in doc.get("workloads", []):
            if pt.get("workload_key") != [read_len, reads_per_step, preset, genome_scale]:
                continue
            have = pt.get("kernel_source_hash")
            if have != cur:
                reason = "%s was collected with kernel sources %s, the library is built from %s: re-run tools/collect_profiles.sh" % (
                    src, have or "(unrecorded)", cur)
                continue
            return {"traffic": pt["bytes_per_launch"].get(stage), "valu": pt.get("valu_wave_insts_per_launch", {}).get(stage),
                    "source": src, "refused": None, "per_kernel": pt.get("per_kernel") or {}}
    return {"traffic": None, "valu": None, "source": None, "refused": reason, "per_kernel": {}}


# kernel groups of tools/pmc_summarize.py -> family index of ma_debug_dp_family_stats (ksw_launch.h: g_dp_family)
DP_FAMILIES = {"k_ksw_ext<1>": 0, "k_ksw_ext<2>": 1, "k_ksw_grp<2>": 2, "k_ksw_grp<4>": 3, "k_ksw_band": 4, "k_ksw_pk": 5, "k_ksw (LDS)": 6,
               "k_ksw_band (long)": 7}


def per_kernel_roofline(per_kernel, fam_cells_per_step):
    """Every DP kernel family and k_seed on its own: duration, VALU instructions and HBM bytes per step from the committed rocprofv3 passes
    (kernel trace + separate PMC passes of the single-stream leg: tools/collect_profiles.sh), the family's cells per step counted live in
    this run (ma_debug_dp_family_stats).  valu_frac: against the chip's issue peak; hbm_frac: against 8 TB/s."""
    out = {}
    for g, o in sorted(per_kernel.items()):
        if not (g.startswith("k_ksw") or g == "k_seed"):
            continue
        ms = o.get("ms_per_step")
        if not ms:
            continue
        e = {"ms": round(ms, 3)}
        if o.get("SQ_INSTS_VALU") is not None:
            e["wave_insts"] = int(o["SQ_INSTS_VALU"])
            e["valu_frac"] = round(o["SQ_INSTS_VALU"] / (ms / 1e3) / 1e9 / CHIP_VALU_PEAK_GINST, 3)
        if o.get("hbm_bytes_per_step") is not None:
            e["hbm_bytes"] = int(o["hbm_bytes_per_step"])
            e["hbm_frac"] = round(o["hbm_bytes_per_step"] / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
        f = DP_FAMILIES.get(g)
        if f is not None and fam_cells_per_step is not None and fam_cells_per_step[2 * f] > 0 and "wave_insts" in e:
            e["cells"] = int(fam_cells_per_step[2 * f])
            e["jobs"] = int(fam_cells_per_step[2 * f + 1])
            e["lane_insts_per_cell"] = round(e["wave_insts"] * 64.0 / e["cells"], 1)
        out[g] = e
    return out


def load_calibration():
    """The measured ceilings the roofline fractions are quoted against (tools/calibrate.sh): the newest
    profiles/rNN_calibration.json; its `kernel_source_hash` (if any) says which build's opcode mix was measured."""
    import glob
    cal = {"gather_ceiling_gblocks": 50.59, "valu_mix_peak_ginst": 576.9, "source": "built-in (no profiles/r*_calibration.json)"}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_calibration.json")))
    if files:
        try:
            with open(files[-1]) as f:
                cal.update(json.load(f))
            cal["source"] = "profiles/" + os.path.basename(files[-1]) + (
                " (measured on kernel sources %s)" % cal["kernel_source_hash"] if cal.get("kernel_source_hash") else "")
        except (OSError, ValueError):
            pass
    return cal


class Env:
    """Process-wide state: device, library, distributed group, genome index."""

    def __init__(self, args):
        import torch
        import ma_amd
        self.torch, self.ma = torch, ma_amd
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        # test hook: MA_BENCH_ONE_DEVICE=1 runs all ranks on GPU 0 over gloo, so that the multi-rank code path (barriers,
        # MAX/SUM reductions, rank-0 reporting) can be exercised on a one-GPU box; the driver's runs use RCCL, one GPU per rank
        self.one_dev = os.environ.get("MA_BENCH_ONE_DEVICE") == "1"
        self.gpu = 0 if self.one_dev else local_rank
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.one_dev:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.gpu))
        torch.cuda.set_device(self.gpu)
        ma_amd.set_device(self.gpu)
        # host threads next to the GPU: the boxes of the pool are two-socket nodes and the scheduler is free to spread this process
        # over both, differently from run to run; MA_BENCH_BIND=none / remote: A/B hooks
        bind = os.environ.get("MA_BENCH_BIND", "local")
        self.cpus_at_start = os.sched_getaffinity(0)  # the CPU baseline (compiled reference, oracle threads) runs on THIS mask
        self.bound_cpus = ma_amd.bind_host_thread(self.gpu, 1 if bind == "remote" else 0) if bind != "none" else 0
        self.bind = bind if self.bound_cpus else "none"
        self.dev = torch.device("cuda", self.gpu)
        self.L = ma_amd.lib()
        self.cal = load_calibration()
        # ---- synthetic genome + index (not timed: one-off preprocessing) -----------------
        self.lens = np.array([max(1000, int(x * args.genome_scale)) for x in GRCH38], dtype=np.uint64)
        self.F = int(self.lens.sum())
        t0 = time.perf_counter()
        g = torch.empty(self.F, dtype=torch.uint8, device=self.dev)
        self.chk(self.L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(self.F), C.c_int32(0 if args.no_repeats else 1),
                                               C.c_void_p(g.data_ptr())))
        self.idx = ma_amd.Index.build_device(self.lens, g.data_ptr())
        del g
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        self.t_index = time.perf_counter() - t0
        self.ref_dir = None  # index files for the compiled reference, written once
        self.live = []  # batch objects of the workload being run
        self.live_host = []  # its page-locked staging arrays

    def chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.ma_last_error().decode())

    def reference_index(self):
        """The GPU-built index in the reference's file formats (read by its own loaders), written once per run."""
        if self.ref_dir is None:
            self.ref_dir = tempfile.mkdtemp(prefix="ma_ref_")
            self.idx.store(os.path.join(self.ref_dir, "idx"))
        return os.path.join(self.ref_dir, "idx")

    def close(self):
        if self.ref_dir:
            shutil.rmtree(self.ref_dir, ignore_errors=True)


def run_workload(E, name, wl, args):
    """Times K steps of one workload; returns its result block (rank 0) incl. roofline, cpu_baseline, parity_check."""
    from ma_amd.shard import reduce_timing_and_counts, shard_range, weak_shard_first_index
    torch, ma_amd, L, dev, dist = E.torch, E.ma, E.L, E.dev, E.dist
    rank, world = E.rank, E.world
    read_len, preset = wl["read_len"], wl.get("preset") or args.preset
    host_io = bool(args.host_io)
    K, W = wl["steps"], wl["warmup"]
    B_total = wl["reads_per_step"]  # reads per step and GPU (weak) / per step over all GPUs (strong)
    if args.scaling == "strong":
        lo, hi = shard_range(B_total, world, rank)
        B = hi - lo  # this rank's share of every step
        first = lambda k: k * B_total + lo  # noqa: E731  global index of this rank's first read of step k
        n_global = B_total * K
    else:
        B = B_total
        first = lambda k: weak_shard_first_index(B_total * K, rank) + k * B  # noqa: E731
        n_global = B_total * K * world
    n_reads = B * K
    cap = int(n_reads * (read_len * (1.0 + 2 * wl["ins"]) + 8)) + 1024
    codes = torch.empty(cap, dtype=torch.uint8, device=dev)
    # reads of step k are the contiguous global indices first(k) .. first(k) + B; every step has its own CSR (B + 1
    # offsets starting at 0, relative to the step's first base) so that a batch takes (codes + base, offsets) as they are
    offs = torch.empty(K * (B + 1) + 1, dtype=torch.int64, device=dev)
    nb_total = 0
    offs_h = np.zeros(n_reads + 1, dtype=np.uint64)
    for k in range(K):
        nb = C.c_uint64()
        E.chk(L.ma_synth_reads_device(E.idx.h, C.c_uint64(wl["seed"]), C.c_uint64(B), C.c_uint32(read_len),
                                      C.c_double(wl["sub"]), C.c_double(wl["ins"]), C.c_double(wl["dele"]),
                                      C.c_uint64(first(k)), C.c_void_p(codes.data_ptr() + nb_total),
                                      C.c_void_p(offs.data_ptr() + 8 * k * (B + 1)), C.c_uint64(cap - nb_total), C.byref(nb)))
        o = offs[k * (B + 1):k * (B + 1) + B + 1].cpu().numpy().astype(np.uint64)
        offs_h[k * B:k * B + B + 1] = o + np.uint64(nb_total)
        nb_total += int(nb.value)
    P = ma_amd.Params.preset(preset)
    max_bases = int((offs_h[B::B] - offs_h[:-1:B]).max()) if K > 0 and B > 0 else 0
    NB = max(1, min(args.inflight, K))
    batches = []

    # ---- host-to-host leg: the reads of every step start in page-locked host memory and the flat
    # result arrays end in page-locked host memory.  The staging holds the reads of up to `hs` distinct steps (step k uses
    # slot k mod hs; at most ~6 GB are page-locked), every batch in flight owns one set of result arrays.
    hcodes = hoffs = None
    hout = []
    hs = 0
    if host_io and n_reads > 0:
        hs = int(max(1, min(K, 6e9 // max(max_bases, 1))))
        hstride = max_bases + 64
        hcodes = ma_amd.HostArray(hs * hstride, np.uint8)
        hoffs = ma_amd.HostArray(hs * (B + 1), np.uint64)
        for j in range(hs):
            a, z = int(offs_h[j * B]), int(offs_h[j * B + B])
            torch.from_numpy(hcodes.a[j * hstride:j * hstride + (z - a)]).copy_(codes[a:z])
            hoffs.a[j * (B + 1):(j + 1) * (B + 1)] = offs_h[j * B:j * B + B + 1] - offs_h[j * B]
        E.live_host += [hcodes, hoffs]

    def grow_out(i, c):
        for h in hout[i] or ():
            h.close()
        hout[i] = (ma_amd.HostArray(B + 1, np.uint64),
                   ma_amd.HostArray(int(c["alignments"] * 1.25) + 4096, ma_amd.ALIGNMENT_DT),
                   ma_amd.HostArray(int(2 * c["ops_cap"] * 1.25) + 4096, np.uint64))

    h2h_only = os.environ.get("MA_BENCH_H2H", "")
    phases = [np.zeros(4) for _ in range(NB)]  # host-to-host leg: seconds in upload / align + wait / download, steps

    def step(i, k):
        bt = batches[i][0]
        lo_r = k * B
        if host_io:
            j = k % hs
            t_a = time.perf_counter()
            if h2h_only == "down":  # diagnostic (MA_BENCH_H2H=down): reads from HBM, results into host memory
                bt.set_reads_device(codes.data_ptr() + int(offs_h[lo_r]), offs.data_ptr() + 8 * k * (B + 1), B, int(offs_h[lo_r + B] - offs_h[lo_r]))
            else:
                bt.set_reads_flat(hcodes.ptr + j * (max_bases + 64), hoffs.ptr + 8 * j * (B + 1), B)
            t_b = time.perf_counter()
            bt.align()
            bt.sync()
            t_c = time.perf_counter()
            if h2h_only != "up":  # diagnostic (MA_BENCH_H2H=up): reads from host memory, results stay in HBM
                if hout[i] is None or bt.mapq_alignments_into(*hout[i]) is None:
                    grow_out(i, bt.counts())  # first step of this batch object, or a step with more output than any before
                    if bt.mapq_alignments_into(*hout[i]) is None:
                        raise RuntimeError("result arrays too small after growing them")
            t_d = time.perf_counter()
            phases[i] += np.array([t_b - t_a, t_c - t_b, t_d - t_c, 1.0])
            return
        nbases = int(offs_h[lo_r + B] - offs_h[lo_r])
        bt.set_reads_device(codes.data_ptr() + int(offs_h[lo_r]), offs.data_ptr() + 8 * k * (B + 1), B, nbases)
        bt.align()
        bt.sync()

    # Waits.  Every batch thread waits for its stream; the runtime's default wait SPINS.  With more waiting threads on the host
    # than the container's CPU quota grants cores (8 ranks x 3 batches in flight = 24 spinners under a 16-core CFS quota) the
    # spinners burn the quota and the whole process group is throttled (that is what bounded the per-read
    # funnel); then the waits sleep on an interrupt-driven event instead (ma_batch_set_blocking_sync).  MA_BENCH_BLOCKING_SYNC=0/1 forces it.
    quota = cpu_quota_cores()
    cores = min(os.cpu_count() or 1, quota) if quota else (os.cpu_count() or 1)
    blocking = world * NB > cores
    if os.environ.get("MA_BENCH_BLOCKING_SYNC") in ("0", "1"):
        blocking = os.environ["MA_BENCH_BLOCKING_SYNC"] == "1"
    setup_err = None
    try:
        for i in range(NB):
            bt = ma_amd.Batch(E.idx, P, max(B, 1), max_bases + 64)
            st = torch.cuda.current_stream() if NB == 1 else torch.cuda.Stream()
            bt.set_stream(st.cuda_stream)
            bt.set_blocking_sync(blocking)
            bt.enable_timing(True)
            batches.append((bt, st))
            hout.append(None)
            E.live.append(bt)  # closed by the caller if this workload fails half-way
        for w in range(W):  # warm-up: W steps on every batch object (each sizes its own buffers); untimed
            for i in range(NB):
                step(i, (w + i) % K)
    except RuntimeError as e:  # e.g. not enough HBM for that many batches in flight
        setup_err = e
    # a rank that failed to set up must not leave the others waiting at the barrier: agree on the outcome first
    failed = torch.tensor([0.0 if setup_err is None else 1.0], dtype=torch.float64, device=torch.device("cpu") if E.one_dev else dev)
    if dist is not None:
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
    if float(failed.item()) > 0:
        raise RuntimeError("setting up %d batch(es) of workload %s failed on %s: %s" % (
            NB, name, "this rank" if setup_err is not None else "another rank", setup_err))

    acc = [dict(kms=np.zeros(8), ctr=np.zeros(8), segs=0, aligned=0, err=None, wall=[], last=None) for _ in range(NB)]

    # host-to-host leg, double-buffered (ma_batch_stage_reads ...): the reads of a batch object's NEXT step are uploaded and the
    # results of its LAST step downloaded on the object's I/O stream while its kernels run; MA_BENCH_H2H_SERIAL=1: upload, kernels
    # and download of a batch object one after the other (ma_batch_set_reads / ma_batch_get_mapq_alignments), the form of rounds 3-4
    pipelined = bool(host_io) and n_reads > 0 and not h2h_only and os.environ.get("MA_BENCH_H2H_SERIAL") != "1"

    def stage(i, k):
        j = k % hs
        batches[i][0].stage_reads_flat(hcodes.ptr + j * (max_bases + 64), hoffs.ptr + 8 * j * (B + 1), B)

    def step_pipelined(i, k, k_next):
        bt = batches[i][0]
        t_a = time.perf_counter()
        bt.use_staged_reads()
        # long reads: the next reads (GBs) are staged AFTER this batch's seeding stage, beside its chaining and DP stages (k_seed_long
        # lives on random gathers and is the kernel a concurrent 2 GB copy disturbs most); MA_BENCH_STAGE_EARLY=1: at the start of the
        # step, as in round 5.  (Ordering uploads against the seeding kernels of OTHER batches by events was built and measured in
        # round 6 -- no gain, profiles/r06_h2h_inflight.txt -- and taken out again.)
        late = read_len > 1000 and k_next is not None and os.environ.get("MA_BENCH_STAGE_EARLY") != "1"
        if k_next is not None and not late:
            stage(i, k_next)
        t_b = time.perf_counter()
        if late:
            bt.seed()
            stage(i, k_next)
            bt.extract()
            bt.chain()
            bt.dp()
        else:
            bt.align()
        bt.sync()
        t_c = time.perf_counter()
        bt.finish_download()  # of the step before (its arrays are re-used)
        if hout[i] is None or bt.start_mapq_download(*hout[i]) is None:
            grow_out(i, bt.counts())
            if bt.start_mapq_download(*hout[i]) is None:
                raise RuntimeError("result arrays too small after growing them")
        t_d = time.perf_counter()
        phases[i] += np.array([t_b - t_a, t_c - t_b, t_d - t_c, 1.0])

    def worker(i):
        try:
            torch.cuda.set_device(E.gpu)
            a = acc[i]
            mine = list(range(i, K, NB))
            if pipelined and mine:
                stage(i, mine[0])
            for at, k in enumerate(mine):
                tk = time.perf_counter()
                if pipelined:
                    step_pipelined(i, k, mine[at + 1] if at + 1 < len(mine) else None)
                else:
                    step(i, k)
                bt = batches[i][0]
                km = bt.kernel_ms().astype(np.float64)
                a["wall"].append((time.perf_counter() - tk) * 1e3)
                if os.environ.get("MA_BENCH_VERBOSE"):
                    print("%s step %d wall %.1f ms, stage ms %s" % (name, k, (time.perf_counter() - tk) * 1e3,
                                                                    np.round(km[:6], 2).tolist()), file=sys.stderr, flush=True)
                a["kms"] += km
                a["ctr"] += bt.counters().astype(np.float64)
                c = bt.counts()
                a["aligned"] += c["aligned_reads"]
                a["segs"] += c["segments"]
                a["last"] = k
            if pipelined:
                batches[i][0].finish_download()
        except Exception as e:  # surfaced after the join
            acc[i]["err"] = e

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    for ph in phases:
        ph[:] = 0
    thr0 = cfs_throttle()
    band0 = band_stats(E)
    bandl0 = band_long_stats(E)
    fam0 = dp_family_stats(E)
    t0 = time.perf_counter()
    if NB == 1:
        worker(0)
    else:
        th = [threading.Thread(target=worker, args=(i,)) for i in range(NB)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    fam1 = dp_family_stats(E)
    thr1 = cfs_throttle()
    for a in acc:
        if a["err"] is not None:
            raise a["err"]
    # ---- the results of a leg with several batches in flight (and of the host-to-host leg) against the same steps run ALONE,
    # one batch at a time, device resident -- after the timed region.  The single-stream path is what the oracle check of
    # cpu_baseline_and_parity covers; this ties the overlapped / host-to-host records to it byte for byte (a race between
    # concurrent batches -- shared DP scratch, result arrays re-allocated mid-run -- would show here).
    xleg = None
    if (NB > 1 or host_io) and n_reads > 0 and not os.environ.get("MA_BENCH_NO_XLEG") and not h2h_only:
        xleg = cross_leg_parity(batches, hout, acc, host_io, hs, B, K, codes, offs, offs_h)
    kms = sum(a["kms"] for a in acc)
    walls = [w for a in acc for w in a["wall"]]
    ctr = sum(a["ctr"] for a in acc)
    segs = sum(a["segs"] for a in acc)
    aligned = sum(a["aligned"] for a in acc)
    dt, (aligned_all,) = reduce_timing_and_counts(dist, torch.device("cpu") if E.one_dev else dev, dt, [aligned])

    # ---- roofline of the dominant kernel (HIP events on the launch stream, averaged over the K launches) --
    total_bases = float(offs_h[n_reads] - offs_h[0])
    alg = [64.0 * ctr[1] + total_bases + 40.0 * segs,  # seeding: occ blocks + read bases + segments out
           64.0 * ctr[2] + 8.0 * ctr[3] + 48.0 * ctr[3],  # SA lookup: LF blocks + SA sample + seed out
           0.0, 0.0,
           ctr[6] + ctr[4] + ctr[7],  # DP: sequences in + 1 B per band cell + back-trace reads + cigar out
           0.0]
    Kd = max(K, 1)
    dom = int(np.argmax(kms[:6]))
    avg_s = kms[dom] / 1e3 / Kd
    ach = (alg[dom] / Kd) / avg_s / 1e9 if avg_s > 0 else 0.0
    # PMC-derived quantities are NOT measured in this process: they come from the committed rocprofv3 passes of the same
    # workload (tools/collect_profiles.sh) and carry their source; null when no pass of this workload is committed
    pmc = pmc_replay(read_len, B_total, preset, args.genome_scale, STAGES[dom])
    traffic, valu, src = pmc["traffic"], pmc["valu"], pmc["source"]
    hbm = {"achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
           "traffic": traffic, "traffic_source": src if traffic is not None else None,
           "algorithmic_bytes_per_launch": int(alg[dom] / Kd)}
    if pmc["refused"]:
        hbm["pmc_replay_refused"] = pmc["refused"]
    roofline = {"kernel": STAGES[dom], "avg_launch_ms": round(kms[dom] / Kd, 3)}
    if dom == 4 and valu and avg_s > 0:
        # the DP kernels are bound by VALU issue, not by HBM: wave-level VALU instructions of the committed PMC pass over
        # the live launch time, against the MEASURED issue rate of the kernel's instruction mix (tools/valu_mix.hip)
        # `peak` = what the chip can issue at all (full-rate ops: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction, MI355X_MICROARCH.md);
        # the measured issue rate of THIS kernel's opcode mix (packed 16-bit ops and DPP moves issue at ~4 cycles: tools/valu_mix.hip,
        # re-run by tools/calibrate.sh) is the secondary `mix_ceiling`
        mix = float(E.cal["valu_mix_peak_ginst"])
        roofline.update({"bound": "valu", "achieved": round(valu / avg_s / 1e9, 1), "peak": CHIP_VALU_PEAK_GINST, "unit": "G wave-inst/s",
                         "frac": round(valu / avg_s / 1e9 / CHIP_VALU_PEAK_GINST, 3), "traffic": traffic,
                         "frac_of_chip_valu_peak": round(valu / avg_s / 1e9 / CHIP_VALU_PEAK_GINST, 3),
                         "mix_ceiling": {"peak": mix, "frac": round(valu / avg_s / 1e9 / mix, 3),
                                         "source": E.cal.get("source", "profiles/r02_valu_mix.txt (dp_mix, 8 waves/SIMD)")},
                         "lane_insts_per_cell": round(valu * 64.0 / max(ctr[4] / Kd, 1.0), 1),
                         "wave_insts_per_launch": valu, "wave_insts_source": src, "hbm": hbm})
    else:
        roofline.update({"bound": "hbm"})
        roofline.update(hbm)
    roofline.update({
        "kernel_ms_per_step": {STAGES[i]: round(kms[i] / Kd, 3) for i in range(6)},
        "seeding_GBps": round((alg[0] / Kd) / (kms[0] / 1e3 / Kd) / 1e9, 2) if kms[0] > 0 else 0.0,
        "seeding_frac_of_gather_ceiling": round(ctr[1] / (kms[0] / 1e3) / 1e9 / float(E.cal["gather_ceiling_gblocks"]), 4) if kms[0] > 0 else 0.0,
        "gather_ceiling_Gblocks_s": float(E.cal["gather_ceiling_gblocks"]),
        "dp_GCUPS": round(ctr[4] / (kms[4] / 1e3) / 1e9, 3) if kms[4] > 0 else 0.0,
        "dp_band_cells_per_read_executed": round(ctr[4] / max(n_reads, 1), 1)})
    pk = per_kernel_roofline(pmc.get("per_kernel") or {}, (fam1 - fam0) / Kd if fam0 is not None and fam1 is not None else None)
    if pk:
        roofline["per_kernel"] = pk
        roofline["per_kernel_source"] = src
    band1 = band_stats(E)
    if band0 is not None and band1 is not None and band1[0] > band0[0]:
        bd = band1 - band0
        # ksw_band.h: extension jobs computed on a band of 24 cells, four per wavefront, each PROVEN afterwards to be the wide band's
        # result; the others go on to the extension kernels.  (The timed region plus the parity / statistics read-backs behind it.)
        roofline["narrow_band"] = {"jobs_tried": int(bd[0]), "proved": int(bd[1]), "failed_check_1_2_3_4": [int(x) for x in bd[2:6]],
                                   "proved_frac": round(float(bd[1]) / float(bd[0]), 4)}

    bandl1 = band_long_stats(E)
    if bandl0 is not None and bandl1 is not None and bandl1[0] > bandl0[0]:
        bd = bandl1 - bandl0
        # ksw_band.h, G = 1: extension jobs of more than 254 query bases (the end extensions of long reads) one per wavefront on a band
        # of 120 cells, proved afterwards or handed on to the exact kernels
        roofline["narrow_band_long"] = {"jobs_tried": int(bd[0]), "proved": int(bd[1]), "failed_check_1_2_3_4": [int(x) for x in bd[2:6]],
                                        "handed_on_otherwise": int(bd[6]), "proved_frac": round(float(bd[1]) / float(bd[0]), 4),
                                        "diagonals_per_job": round(float(bd[7]) / float(bd[0]), 1)}

    # ---- CPU baseline (rank 0, N = 1): the compiled reference and the oracle on this host's cores, then parity ---------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample != 0 and n_reads > 0 and not host_io:
        pinned = os.sched_getaffinity(0)
        try:  # the host threads were pinned next to the GPU for the GPU legs (one socket); the CPU baseline gets every CPU the process had
            os.sched_setaffinity(0, E.cpus_at_start)
            cpu = cpu_baseline_and_parity(E, name, wl, args, codes, offs_h, B, n_reads, step, batches, roofline)
        finally:
            os.sched_setaffinity(0, pinned)

    res = None
    if rank == 0:
        res = {
            "name": name, "baseline_config": wl.get("baseline_config"), "batches_in_flight": NB,
            "io": ("host to host (reads from page-locked host memory, flat results into page-locked host memory%s)" % (
                       "; a batch object's next upload and last download run beside its kernels" if pipelined else "")) if host_io
                  else "device resident (reads and results stay in HBM)",
            "workload": "%d x %d bp reads (%.2f%% sub, %.2f%% ins, %.2f%% del), %s preset, %d reads/step %s" % (
                n_global, read_len, 100 * wl["sub"], 100 * wl["ins"], 100 * wl["dele"], preset, B_total,
                "per GPU" if args.scaling == "weak" else "over all GPUs"),
            "value": round(aligned_all / dt, 1), "unit": "aligned reads/s", "aligned_reads": int(aligned_all), "steps": K, "warmup": W,
            "ms_per_step": round(dt / Kd * 1e3, 3),
            # wall time of the single steps on this rank (with several batches in flight: of a step on its own stream)
            "step_ms_min": round(min(walls), 3) if walls else None, "step_ms_max": round(max(walls), 3) if walls else None,
            "reads_per_s_total": round(n_global / dt, 1),
            "gbases_per_s": round(total_bases * (world if args.scaling == "weak" else 1) / dt / 1e9, 3),
            "roofline": roofline, "cpu_baseline": cpu,
            "stream_waits": "blocking (event)" if blocking else "spinning (runtime default)",
            "cfs_throttled": None if thr0 is None or thr1 is None else {"periods": thr1[0] - thr0[0], "thread_seconds": round(thr1[1] - thr0[1], 3)},
            "host_cores": cores, "cross_leg_parity": xleg,
            "host_threads": ("pinned to the %d CPUs %s the GPU (ma_host_bind_thread)" % (E.bound_cpus, "next to" if E.bind == "local" else "AWAY from")
                             if E.bound_cpus else "not pinned"),
        }
        if host_io:
            ph = sum(phases)
            if ph[3] > 0:  # wall time of a batch thread per step, by phase (the threads of the other batches run meanwhile)
                res["host_phases_ms_per_step"] = {"upload": round(ph[0] / ph[3] * 1e3, 3), "align_and_wait": round(ph[1] / ph[3] * 1e3, 3),
                                                  "download": round(ph[2] / ph[3] * 1e3, 3)}
            if h2h_only:
                res["io"] += "; MA_BENCH_H2H=%s (diagnostic: one direction only)" % h2h_only
        if host_io and hs < K:
            res["io"] += "; the page-locked staging holds the reads of %d steps: step k re-uses the reads of step k mod %d" % (hs, hs)
    for bt, _ in batches:
        bt.close()
    E.live.clear()
    for hset in hout:
        for h in hset or ():
            h.close()
    for h in (hcodes, hoffs):
        if h is not None:
            h.close()
    E.live_host = []
    del codes, offs
    torch.cuda.empty_cache()
    return res


def cross_leg_parity(batches, hout, acc, host_io, hs, B, K, codes, offs, offs_h):
    """The MappingQuality records (offsets, every header byte, every op) of the LAST step each batch object ran inside the timed
    region -- with the other batches in flight beside it, host to host or device resident -- against the same reads aligned
    again with nothing else running (device resident, on batch object 0)."""
    got = []
    for i, (bt, _) in enumerate(batches):
        k = acc[i]["last"]
        if k is None:
            continue
        if host_io:
            ho, ha, hp = hout[i]
            off = ho.a[:B + 1].copy()
            na = int(off[B])
            al = ha.a[:na].copy()
            nops = int(al["ops_off"][na - 1] + al["n_ops"][na - 1]) if na else 0
            ops = hp.a[:2 * nops].copy()
            k_reads = k % hs
        else:
            off, al, ops = bt.mapq_alignments()
            na = int(off[B])
            al = al[:na]
            nops = int(al["ops_off"][na - 1] + al["n_ops"][na - 1]) if na else 0
            ops = ops[:2 * nops]
            k_reads = k
        got.append((k, k_reads, off, al, ops, na, nops))
    bad_steps, n_al, n_ops = 0, 0, 0
    bt0 = batches[0][0]
    for k, kr, off, al, ops, na, nops in got:
        lo_r = kr * B
        bt0.set_reads_device(codes.data_ptr() + int(offs_h[lo_r]), offs.data_ptr() + 8 * kr * (B + 1), B, int(offs_h[lo_r + B] - offs_h[lo_r]))
        bt0.align()
        bt0.sync()
        woff, wal, wops = bt0.mapq_alignments()
        same = (np.array_equal(off[:B + 1], woff[:B + 1]) and al.tobytes() == wal[:na].tobytes()
                and np.array_equal(ops, wops[:2 * nops]))
        bad_steps += 0 if same else 1
        n_al += na
        n_ops += nops
    return {"steps_compared": len(got), "reads": len(got) * B, "alignments": n_al, "ops": n_ops, "mismatching_steps": bad_steps,
            "what": "the last step of every batch in flight vs the same reads aligned alone (single stream, device resident): MappingQuality "
                    "records byte for byte"}


def cpu_baseline_and_parity(E, name, wl, args, codes, offs_h, B, n_reads, step, batches, roofline):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ma_testlib import OrIndex, or_params, write_case
    ncores = os.cpu_count() or 1
    read_len = wl["read_len"]
    S = args.cpu_sample if args.cpu_sample > 0 else (wl["cpu_sample"] or min(n_reads, max(2000, int(4000 * ncores * 150 / max(read_len, 1)))))
    S = min(S, n_reads)
    preset = wl.get("preset") or args.preset
    oidx = OrIndex.from_parts(E.idx.download())
    hb = int(offs_h[S])
    rc = codes[:hb].cpu().numpy()
    reads = [rc[int(offs_h[i]):int(offs_h[i + 1])] for i in range(S)]
    op = or_params(preset, 1)
    t1 = time.perf_counter()
    res = oidx.align(reads, op, threads=ncores)
    tc = time.perf_counter() - t1
    cpu = {"value": round(res["n_aligned"] / tc, 1), "unit": "aligned reads/s", "cores": ncores, "kind": "port",
           "sample": "first %d reads of the same workload, oracle with %d threads, %.1f s" % (S, ncores, tc),
           # the reference computes every diagonal of an extension; the GPU path stops once ez.max is final
           "dp_band_cells_per_read": round(float(res["counters"][4]) / S, 1)}
    # ---- the REAL reference (compiled from its own sources into oracle/_ref by oracle/Makefile.ref) on the same
    # sample: the GPU-built index is written in the reference's file formats, the reference's own loaders read it,
    # its modules (BinarySeeding .. MappingQuality) run on the host threads (oracle/ref_dump.cpp timeidx).  The reference
    # scales badly over many threads (glibc rand() lock, allocator), so a few thread counts are timed and the best is
    # the baseline; the 1-thread rate is measured on a smaller sample.
    ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if os.path.exists(ref_dump) and not os.environ.get("MA_BENCH_NO_REFERENCE") and wl.get("reference", True):
        td = tempfile.mkdtemp(prefix="ma_reads_")
        try:
            t2 = time.perf_counter()
            prefix = E.reference_index()
            t_store = time.perf_counter() - t2
            write_case(os.path.join(td, "reads.case"), [], reads)
            S1 = max(4, min(S // 64, int(1e6 / max(read_len, 1))))
            write_case(os.path.join(td, "reads1.case"), [], reads[:S1])

            def timed(case, threads):
                out = subprocess.run([ref_dump, "timeidx", prefix, os.path.join(td, case), preset, str(threads)],
                                     capture_output=True, text=True, timeout=1800)
                m = re.search(r"(\d+) reads \((\d+) aligned\) in ([0-9.]+) s on (\d+) threads", out.stdout)
                if out.returncode != 0 or not m:
                    raise RuntimeError((out.stderr or out.stdout)[-300:])
                return int(m.group(2)) / float(m.group(3)), float(m.group(3))

            runs = {}
            quota = cpu_quota_cores()
            cand = sorted(set([ncores] + ([max(1, ncores // 4)] if args.cpu_threads_sweep else [])
                              + ([max(1, int(quota))] if quota and quota < ncores else [])))  # as many threads as the quota grants cores
            light = bool(wl.get("preset"))  # the legs under another parameter set: ONE run, with the threads the quota grants (every run
            if light:                       # of ref_dump loads the GRCh38-size index again: the default bench line stays within minutes)
                cand = [max(1, int(quota))] if quota and quota < ncores else [ncores]
            for t in cand:
                runs[t] = timed("reads.case", t)
            best = max(runs, key=lambda t: runs[t][0])
            one = (None, None) if light else timed("reads1.case", 1)
            cpu = dict(cpu, value=round(runs[best][0], 1), cores=best, kind="reference",
                       sample="first %d reads of the same workload, the reference's own modules (BinarySeeding .. "
                              "MappingQuality), %.1f s on %d threads; index written by the GPU builder and loaded by the "
                              "reference's loaders (%.0f s to write, not counted)" % (S, runs[best][1], best, t_store),
                       by_threads={str(t): round(v[0], 1) for t, v in runs.items()},
                       one_thread=None if one[0] is None else {"value": round(one[0], 1), "sample": "first %d reads, %.1f s" % (S1, one[1])},
                       port={"value": cpu["value"], "sample": cpu["sample"]})
        except Exception as e:  # the oracle's number stays
            cpu["reference_error"] = repr(e)[:300]
        finally:
            shutil.rmtree(td, ignore_errors=True)
    # ---- parity at full scale: the GPU results of the sampled reads of step 0 against the oracle's, bit for bit
    # (NeedlemanWunsch output incl. every alignment op, and the MappingQuality records incl. mapq doubles)
    Pn = min(S, B)
    step(0, 0)
    bt = batches[0][0]
    goff, galn, gops = bt.alignments()
    moff, mq, _ = bt.mapq_alignments()
    na, nm = int(res["aln_off"][Pn]), int(res["mq_off"][Pn])
    nops = int(res["alns"]["ops_off"][na - 1] + res["alns"]["n_ops"][na - 1]) if na else 0
    same = (np.array_equal(goff[:Pn + 1], res["aln_off"][:Pn + 1]) and int(goff[Pn]) == na
            and galn[:na].tobytes() == res["alns"][:na].tobytes()
            and np.array_equal(gops[:2 * nops], res["ops"][:2 * nops])
            and np.array_equal(moff[:Pn + 1], res["mq_off"][:Pn + 1]))
    if same:
        for f in ("begin_ref", "end_ref", "begin_q", "end_q", "score", "soc_index", "n_ops", "secondary", "supplementary"):
            same = same and np.array_equal(mq[f][:nm], res["mq"][f][:nm])
        same = same and mq["mapq"][:nm].tobytes() == res["mq"]["mapq"][:nm].tobytes()
    bad = 0
    if not same:  # count the reads that differ
        for r in range(Pn):
            a0, a1 = int(res["aln_off"][r]), int(res["aln_off"][r + 1])
            g0, g1 = int(goff[r]), int(goff[r + 1])
            ok = (a1 - a0) == (g1 - g0)
            if ok:
                for f in ("begin_ref", "end_ref", "begin_q", "end_q", "score", "soc_index", "n_ops"):
                    ok = ok and np.array_equal(galn[f][g0:g1], res["alns"][f][a0:a1])
                for k in range(a1 - a0):
                    oa, ga = res["alns"][a0 + k], galn[g0 + k]
                    ok = ok and np.array_equal(gops[2 * int(ga["ops_off"]):2 * int(ga["ops_off"] + ga["n_ops"])],
                                               res["ops"][2 * int(oa["ops_off"]):2 * int(oa["ops_off"] + oa["n_ops"])])
            bad += 0 if ok else 1
        bad = max(bad, 1)
    cpu["cpu_quota_cores"] = cpu_quota_cores()  # what `cores` threads actually get: the container's CFS quota
    cpu["parity_check"] = {"reads": Pn, "alignments": na, "alignment_ops": nops, "mapq_records": nm,
                           "mismatching_reads": bad,
                           "what": "GPU vs oracle on the first reads of step 0: every NeedlemanWunsch alignment (positions, score, "
                                   "ops) and MappingQuality record (flags, mapq bits)"}
    return cpu


def band_stats(E):
    """ma_debug_band_stats: jobs tried / proved on the narrow band since the library was loaded (this process, this device)"""
    try:
        out = (C.c_ulonglong * 8)()
        if E.L.ma_debug_band_stats(out) != 0:
            return None
        return np.array(list(out), dtype=np.int64)
    except Exception:  # noqa: BLE001
        return None


def dp_family_stats(E):
    """ma_debug_dp_family_stats: cells / jobs per DP kernel family since the library was loaded (this process, this device)"""
    try:
        out = (C.c_ulonglong * 16)()
        if E.L.ma_debug_dp_family_stats(out) != 0:
            return None
        return np.array(list(out), dtype=np.float64)
    except Exception:  # noqa: BLE001
        return None


def band_long_stats(E):
    """ma_debug_band_long_stats: the long extension jobs, one per wavefront on the band of 120"""
    try:
        out = (C.c_ulonglong * 8)()
        if E.L.ma_debug_band_long_stats(out) != 0:
            return None
        return np.array(list(out), dtype=np.int64)
    except Exception:  # noqa: BLE001
        return None


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup CFS quota), or None: the boxes of the pool show 256 logical CPUs and
    grant 16 cores' worth of time per 100 ms period -- every host-side figure of this line is measured under that quota."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if q <= 0 else round(q / p, 2)
        except Exception:  # noqa: BLE001
            return None


def cfs_throttle():
    """(periods in which the container was throttled, seconds its threads spent throttled) so far, or None"""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines() if len(l.split()) == 2)
        return int(d["nr_throttled"]), int(d["throttled_usec"]) / 1e6
    except Exception:  # noqa: BLE001
        return None


def boundary_leg(E, args):
    """Host-fed end-to-end rate through the drop-in boundary (reads in host memory -> BatchAligner -> Alignment
    containers -> SAM text), examples/ma_boundary_bench.cpp; reported in config.boundary, never in `value`."""
    exe = os.path.join(ROOT, "examples", "ma_boundary_bench")
    if not os.path.exists(exe):
        return {"error": "examples/ma_boundary_bench not built (run __graft_entry__.build())"}
    try:
        t0 = cfs_throttle()
        out = subprocess.run([exe, E.reference_index(), str(args.boundary_reads), "150", args.preset, str(E.gpu)],
                             capture_output=True, text=True, timeout=900)
        t1 = cfs_throttle()
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            return {"error": (out.stderr or out.stdout)[-300:]}
        res = json.loads(line[-1])
        gl = [l for l in out.stderr.splitlines() if l.startswith("graph leg:")]
        if gl and isinstance(res.get("graph"), dict):
            res["graph"]["device_side"] = gl[-1]  # wall of the leg vs the seconds its device batches took, by phase
        res["host_cpu_quota_cores"] = cpu_quota_cores()
        if t0 and t1:
            res["cfs_throttled"] = {"periods": t1[0] - t0[0], "thread_seconds": round(t1[1] - t0[1], 1),
                                    "what": "CFS throttling of the container while this leg ran: its host threads (256 workers, 2048 "
                                            "graph threads in the per-read leg) compete for the quota above, which makes the "
                                            "thread-heavy legs vary from run to run"}
        return res
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def self_launch_or_check(args, argv):
    """`--gpus N` is the contract: outside a torchrun environment N > 1 ranks are started HERE, as a child (one process per
    GPU, RCCL), before this process has touched a GPU; inside one, WORLD_SIZE must be N (a 1-GPU number must never be printed
    as an N-GPU one).  The reference's counterpart is one graph copy per thread (export.cpp:99-126, module.h:303-369)."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            print("bench.py: --gpus %d but WORLD_SIZE=%s: refusing to report a %s-rank run as %d GPUs" % (args.gpus, ws, ws, args.gpus),
                  file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    if os.environ.get("MA_BENCH_ONE_DEVICE") != "1" and os.environ.get("MA_BENCH_DRY_RUN") != "1":
        import torch  # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < args.gpus:
            print("bench.py: --gpus %d but this node shows %d GPU(s)" % (args.gpus, have), file=sys.stderr)
            sys.exit(2)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.call(cmd))


def dry_run(args):
    """MA_BENCH_DRY_RUN=1 (CPU tests of the launch path): the ranks rendezvous over gloo, agree on who is there and rank 0
    prints the line's launch fields -- no GPU is touched, nothing is measured."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    seen = [rank]
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = 1
        dist.all_reduce(t)
        seen = [i for i in range(world) if int(t[i]) == 1]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": seen, "gpus_flag": args.gpus,
                          "local_rank_env": os.environ.get("LOCAL_RANK")}))


def c1_anchor(E, args):
    """BASELINE.md section 3's sanity anchor on configs[0] (C1): 1 k x 150 bp reads vs `ecoli_like` (one contig of 4 641 652 nt,
    i.i.d., seed 1; the doubled text of 9.28 Mnt lies just below the 10 Mnt switch, so the drop-all / SoC heuristics are OFF,
    binarySeeding.cpp:172-175, stripOfConsideration.cpp:21-23).  The oracle is raced against the compiled reference (one thread
    each, the reference with ITS OWN index builder) and all three -- reference, oracle, GPU path -- must agree record by record."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ma_testlib import OrIndex, or_params, parse_pipe_dump, write_case
    torch, ma_amd, L = E.torch, E.ma, E.L
    F, n, rl = 4641652, 1000, 150
    out = {"config": "configs[0] (C1): %d x %d bp reads vs ecoli_like (%d nt, seed 1), default preset, heuristics off (< 10 Mnt)" % (n, rl, F)}
    g = torch.empty(F, dtype=torch.uint8, device=E.dev)
    E.chk(L.ma_synth_genome_device(C.c_uint64(1), C.c_uint64(F), C.c_int32(0), C.c_void_p(g.data_ptr())))
    idx = ma_amd.Index.build_device(np.array([F], dtype=np.uint64), g.data_ptr())
    contig = g.cpu().numpy()
    del g
    codes = torch.empty(n * (rl + 8) + 1024, dtype=torch.uint8, device=E.dev)
    offs = torch.empty(n + 2, dtype=torch.int64, device=E.dev)
    nb = C.c_uint64()
    E.chk(L.ma_synth_reads_device(idx.h, C.c_uint64(11), C.c_uint64(n), C.c_uint32(rl), C.c_double(0.005), C.c_double(0.0), C.c_double(0.0),
                                  C.c_uint64(0), C.c_void_p(codes.data_ptr()), C.c_void_p(offs.data_ptr()), C.c_uint64(codes.numel()), C.byref(nb)))
    oh = offs[:n + 1].cpu().numpy().astype(np.int64)
    ch = codes.cpu().numpy()
    reads = [ch[oh[i]:oh[i + 1]] for i in range(n)]
    P = ma_amd.Params.preset("default")
    bt = ma_amd.Batch(idx, P, n, int(oh[n]) + 64)
    try:
        bt.set_reads(reads)
        bt.align()
        bt.sync()
        goff, galn, gops = bt.alignments()
        moff, gmq, _ = bt.mapq_alignments()
        # the oracle on the GPU-built index, one thread, best of three
        oidx = OrIndex.from_parts(idx.download())
        op = or_params("default", 1)
        REP = 20  # the race is timed on 20 x the 1 000 reads (a 1 000-read run of the reference takes 25 ms)
        t_or = []
        for _ in range(3):
            t0 = time.perf_counter()
            oidx.align(reads * REP, op, threads=1)
            t_or.append(time.perf_counter() - t0)
        res = oidx.align(reads, op, threads=1)
        na = int(res["aln_off"][n])
        same = (np.array_equal(goff, res["aln_off"][:n + 1]) and galn[:na].tobytes() == res["alns"][:na].tobytes()
                and np.array_equal(moff, res["mq_off"][:n + 1])
                and gmq["mapq"][:int(moff[n])].tobytes() == res["mq"]["mapq"][:int(moff[n])].tobytes())
        out.update({"oracle_reads_per_s_1_thread": round(REP * n / min(t_or), 1), "gpu_vs_oracle_identical": bool(same),
                    "alignments": na, "aligned_reads": int(res["n_aligned"])})
        ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
        if os.path.exists(ref_dump) and not os.environ.get("MA_BENCH_NO_REFERENCE"):
            td = tempfile.mkdtemp(prefix="ma_c1_")
            try:
                case = os.path.join(td, "c1.case")
                write_case(case, [contig], reads)
                case_t = os.path.join(td, "c1x.case")
                write_case(case_t, [contig], reads * REP)
                t_ref = []
                for _ in range(3):  # `time`: the reference builds its own index of the case's contigs, then times its modules
                    o = subprocess.run([ref_dump, "time", case_t, "default", "1"], capture_output=True, text=True, timeout=600)
                    m = re.search(r"(\d+) reads \((\d+) aligned\) in ([0-9.]+) s on (\d+) threads", o.stdout)
                    if o.returncode != 0 or not m:
                        raise RuntimeError((o.stderr or o.stdout)[-300:])
                    t_ref.append(float(m.group(3)))
                subprocess.check_call([ref_dump, "pipe", case, "default", "1", os.path.join(td, "ref.pipe")], stdout=subprocess.DEVNULL,
                                      timeout=600)
                want = parse_pipe_dump(os.path.join(td, "ref.pipe"))
                bad = 0
                for r in range(n):
                    a0, a1 = int(goff[r]), int(goff[r + 1])
                    w = want[r]["alns"]
                    ok = len(w) == a1 - a0
                    for k in range(a1 - a0 if ok else 0):
                        ga, wa = galn[a0 + k], w[k]
                        o0 = int(ga["ops_off"])
                        ok = ok and (int(ga["begin_ref"]), int(ga["end_ref"]), int(ga["begin_q"]), int(ga["end_q"]), int(ga["score"])) == (
                            wa["bref"], wa["eref"], wa["bq"], wa["eq"], wa["score"])
                        ok = ok and [(int(gops[2 * (o0 + j)]), int(gops[2 * (o0 + j) + 1])) for j in range(int(ga["n_ops"]))] == wa["ops"]
                    m0, m1 = int(moff[r]), int(moff[r + 1])
                    ok = ok and len(want[r]["mq"]) == m1 - m0
                    for k in range(m1 - m0 if ok else 0):
                        ok = ok and float("%.17g" % gmq[m0 + k]["mapq"]) == want[r]["mq"][k]["mapq"]
                    bad += 0 if ok else 1
                out.update({"reference_reads_per_s_1_thread": round(REP * n / min(t_ref), 1),
                            "oracle_over_reference": round(min(t_ref) / min(t_or), 3),
                            "gpu_vs_reference_mismatching_reads": bad,
                            "what": "the reference builds its own index of ecoli_like (is_bwt path); its alignments and mapping "
                                    "qualities of all reads are compared with the GPU path's on the GPU-built index"})
            except Exception as e:  # noqa: BLE001
                out["reference_error"] = repr(e)[:300]
            finally:
                shutil.rmtree(td, ignore_errors=True)
    finally:
        bt.close()
        idx.close()
    return out


def run_legs(E, name, wl, args):
    """The three legs of one workload (module docstring); returns the single-stream block with the other legs attached."""
    import copy
    a1 = copy.copy(args)
    a1.inflight, a1.host_io = 1, 0
    r = run_workload(E, name, wl, a1)
    nfl = args.overlap if wl["read_len"] <= 1000 else min(args.overlap, args.overlap_long)
    # batches in flight of the host-to-host leg: a workload may ask for its own number (WORKLOADS[..]["h2h_inflight"]: 10 kb reads run
    # best with ONE batch object whose uploads and downloads run beside its own kernels -- a second batch's seeding crawls beside the
    # first one's persistent DP waves, profiles/r06_h2h_inflight.txt); --overlap 0 / 1 still means: single-stream leg only
    nfl_h2h = min(nfl, wl.get("h2h_inflight", nfl)) if nfl > 1 else nfl
    if nfl > 1 and not wl.get("single_only"):
        for key, hio in (("overlapped", 0), ("host_to_host", 1)):
            nfl_leg = nfl_h2h if hio else nfl
            wl2 = dict(wl)
            wl2["steps"] = max(wl["steps"], 2 * nfl_leg)  # every batch object gets at least two timed steps
            a2 = copy.copy(args)
            a2.inflight, a2.cpu_sample, a2.host_io = nfl_leg, 0, hio
            try:
                r2, err = run_workload(E, name, wl2, a2), None
            except RuntimeError as e:  # e.g. not enough HBM for that many long-read batches
                r2, err = None, str(e)
                for bt in E.live:
                    bt.close()
                E.live.clear()
                for h in E.live_host:
                    h.close()
                E.live_host = []
                E.torch.cuda.empty_cache()
            if r is None:
                continue
            if r2 is None:
                r[key] = {"batches_in_flight": nfl_leg, "error": err}
                continue
            rf2 = dict(r2["roofline"])
            rf2["leg"] = ("%d batches in flight, %s: a kernel's launch time includes the share of the chip the other batches' kernels "
                          "took meanwhile" % (nfl_leg, r2["io"]))
            r[key] = {"batches_in_flight": nfl_leg, "io": r2["io"], "value": r2["value"], "unit": r2["unit"], "steps": r2["steps"],
                      "ms_per_step": r2["ms_per_step"], "step_ms_min": r2["step_ms_min"], "step_ms_max": r2["step_ms_max"],
                      "gbases_per_s": r2["gbases_per_s"], "aligned_reads": r2["aligned_reads"],
                      "kernel_ms_per_step_under_overlap": r2["roofline"]["kernel_ms_per_step"], "roofline": rf2,
                      "cross_leg_parity": r2.get("cross_leg_parity"), "stream_waits": r2.get("stream_waits"),
                      "cfs_throttled": r2.get("cfs_throttled")}
            if r2.get("host_phases_ms_per_step"):
                r[key]["host_phases_ms_per_step"] = r2["host_phases_ms_per_step"]
    return r


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="all", choices=["all", "150bp", "10kb", "50kb", "illumina", "10kb_pacbio", "50kb_nanopore"])
    ap.add_argument("--reads-per-step", type=int, default=0, help="override the workload's reads per step")
    ap.add_argument("--read-len", type=int, default=0, help="custom single workload with --sub/--ins/--dele")
    ap.add_argument("--sub", type=float, default=0.005)
    ap.add_argument("--ins", type=float, default=0.0)
    ap.add_argument("--dele", type=float, default=0.0)
    ap.add_argument("--genome-scale", type=float, default=1.0, help="fraction of GRCh38 contig lengths (tests only)")
    ap.add_argument("--preset", default="default")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1 per workload, 0 off)")
    ap.add_argument("--cpu-threads-sweep", type=int, default=0, help="also time the reference on a quarter of the threads")
    ap.add_argument("--boundary-reads", type=int, default=1000000, help="reads of the host-fed boundary leg (0 off)")
    ap.add_argument("--no-repeats", action="store_true")
    ap.add_argument("--overlap", type=int, default=3, help="batches in flight of the overlapped legs (0/1: single-stream leg only)")
    ap.add_argument("--overlap-long", type=int, default=2, help="the same for reads longer than 1 kb (a batch holds ~110 GB of HBM)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="run ONE leg only, with this many batches in flight per GPU (own stream + host thread each; profiling runs)")
    ap.add_argument("--host-io", type=int, default=0, help="with --inflight: that leg host to host (1) or device resident (0)")
    ap.add_argument("--detail-file", default="", help="where the per-workload blocks go (default: gpurun_out/bench_detail.json)")
    return ap


def main():
    # A batch of long reads runs its DP classes on up to four streams beside its I/O stream, and the runtime maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4): streams that share a queue run one after the other.  After the 150 bp legs of this
    # process had created and dropped a dozen streams, the 10 kb batch's main stream and its k_ksw_pk<5> stream landed on ONE queue and
    # the DP stage took 196 instead of 148 ms (profiles/r06_hw_queues.txt).  Must be set before the first HIP call of the process.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    args = build_parser().parse_args()
    self_launch_or_check(args, sys.argv[1:])
    if os.environ.get("MA_BENCH_DRY_RUN") == "1":
        return dry_run(args)

    E = Env(args)
    wls = []
    if args.read_len > 0:  # custom single workload
        B = args.reads_per_step if args.reads_per_step > 0 else (1000000 if args.read_len <= 1000 else max(10000, int(2e9 / args.read_len)))
        wls.append(("custom", dict(read_len=args.read_len, sub=args.sub, ins=args.ins, dele=args.dele,
                                   seed=11 if args.read_len <= 1000 else (12 if args.read_len <= 20000 else 13),
                                   reads_per_step=B, steps=args.steps, warmup=args.warmup, cpu_sample=None,
                                   baseline_config=None)))
    else:
        names = ["150bp", "10kb", "50kb", "illumina", "10kb_pacbio", "50kb_nanopore"]
        if os.environ.get("MA_BENCH_ONLY"):  # diagnostics: a subset of the default run's workloads, in its order and with its step counts
            names = [n for n in names if n in os.environ["MA_BENCH_ONLY"].split(",")]
        for name in (names if args.workload == "all" else [args.workload]):
            wl = dict(WORKLOADS[name])
            if wl["steps"] is None or args.workload != "all":
                wl["steps"], wl["warmup"] = args.steps, args.warmup
            if args.reads_per_step > 0 and (wl["read_len"] <= 1000 or args.workload != "all"):
                wl["reads_per_step"] = args.reads_per_step
            wls.append((name, wl))
    results = []
    for name, wl in wls:
        if args.inflight > 0:  # one explicit leg (what a rocprofv3 run of a single leg uses)
            r = run_workload(E, name, wl, args)
        else:
            r = run_legs(E, name, wl, args)
        if r is not None:
            results.append(r)
    boundary = anchor = None
    full = E.rank == 0 and E.world == 1 and args.cpu_sample != 0 and args.read_len == 0 and args.inflight == 0
    if full and args.workload == "all":
        try:
            anchor = c1_anchor(E, args)
        except Exception as e:  # noqa: BLE001
            anchor = {"error": repr(e)[:300]}
    if full and args.boundary_reads > 0:
        boundary = boundary_leg(E, args)
    if E.rank == 0:
        print(compose_line(E, args, results, boundary, anchor))
    E.close()
    if E.dist is not None:
        E.dist.destroy_process_group()


def leg_of(block, key):
    """(value, ms_per_step, batches in flight) of a workload's leg, or Nones"""
    leg = block if key is None else (block.get(key) or {})
    return leg.get("value"), leg.get("ms_per_step"), leg.get("batches_in_flight")


def compose_line(E, args, results, boundary, anchor):
    """The ONE line of rank 0: top-level scalars for every workload (the driver keeps the tail of stdout and the scalar fields
    of the line), the roofline and CPU baseline of the leg `value` comes from; everything else goes to the detail file."""
    head = results[0]
    # headline = the host-to-host leg (first read in host memory -> last alignment record in host memory)
    key = "host_to_host" if "value" in (head.get("host_to_host") or {}) else ("overlapped" if "value" in (head.get("overlapped") or {}) else None)
    top = head if key is None else head[key]
    detail = {"workloads": results, "boundary": boundary, "c1_anchor": anchor, "kernel_source_hash": kernel_source_hash()}
    dpath = args.detail_file or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(dpath), exist_ok=True)
        with open(dpath, "w") as f:
            json.dump(detail, f)
    except OSError as e:
        dpath = "not written (%s)" % e
    print("bench detail: " + json.dumps(detail), file=sys.stderr, flush=True)
    out = {
        "metric": "aligned reads/sec (whole node), 150bp & 10kb synthetic vs GRCh38",
        "value": top["value"], "unit": "aligned reads/s", "n_gpus": E.world, "steps": top["steps"], "warmup": head["warmup"],
        "ms_per_step": top["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u8/int8 (2-bit BWT ranks, int8 DP differences, int16/32 scores)",
        "data": "synthetic",
    }
    # every workload as top-level scalars: host to host (the timed region of BASELINE.md section 3), device resident with the same
    # batches in flight, and one batch at a time (the leg the per-kernel roofline is measured on)
    for r in results:
        n = {"150bp": "150bp", "10kb": "10kb", "50kb": "50kb", "illumina": "150bp_illumina"}.get(r["name"], r["name"])  # (+ 10kb_pacbio, 50kb_nanopore)
        v, ms, nb = leg_of(r, "host_to_host")
        if v is not None:
            out["value_%s" % n], out["ms_per_step_%s" % n], out["batches_in_flight_%s" % n] = v, ms, nb
        v2, ms2, nb2 = leg_of(r, "overlapped")
        if v2 is not None:
            out["value_%s_device_resident" % n], out["ms_per_step_%s_device_resident" % n] = v2, ms2
        # what the host-to-host leg loses to its copies: against the device-resident leg with the SAME batches in flight (one batch at a
        # time when the workload's host-to-host leg runs one batch object with its I/O beside its own kernels)
        vd = r.get("value") if nb == 1 and r.get("batches_in_flight", 1) == 1 else (v2 if nb == nb2 else None)
        if v is not None and vd:
            out["h2h_over_device_resident_%s" % n] = round(v / vd, 3)
        for key2, tag in (("host_to_host", "h2h"), ("overlapped", "overlapped")):
            xp = (r.get(key2) or {}).get("cross_leg_parity")
            if xp:
                out["parity_%s_%s" % (n, tag)] = "%d mismatching of %d steps (%d reads, %d alignments, %d ops) vs the same steps run alone" % (
                    xp["mismatching_steps"], xp["steps_compared"], xp["reads"], xp["alignments"], xp["ops"])
        if E.world > 1:  # N > 1: what the container's CPU quota did to the ranks' host threads during the leg `value` comes from
            lg = r.get("host_to_host") or r.get("overlapped") or r
            out["stream_waits_%s" % n] = lg.get("stream_waits")
            out["cfs_throttled_%s" % n] = lg.get("cfs_throttled")
        if r.get("batches_in_flight", 1) == 1 and r.get("io", "device").startswith("device"):
            out["value_%s_single_stream" % n], out["ms_per_step_%s_single_stream" % n] = r["value"], r["ms_per_step"]
        if v is None and v2 is None:
            out["value_%s" % n], out["ms_per_step_%s" % n], out["batches_in_flight_%s" % n] = r["value"], r["ms_per_step"], r.get("batches_in_flight")
        cb = r.get("cpu_baseline") or {}
        if cb.get("value") is not None and cb.get("kind") == "reference":
            out["cpu_reference_%s" % n] = cb["value"]
            out["cpu_reference_threads_%s" % n] = cb.get("cores")
        pc = cb.get("parity_check")
        if pc:
            out["parity_%s" % n] = "%d mismatching of %d reads (%d alignments, %d ops) vs oracle" % (
                pc["mismatching_reads"], pc["reads"], pc["alignments"], pc["alignment_ops"])
        rf = r.get("roofline") or {}
        if rf.get("kernel"):
            out["roofline_%s" % n] = "%s %.2f ms/launch: %s frac %s of %s %s%s; hbm frac %s" % (
                rf["kernel"], rf["avg_launch_ms"], rf.get("bound"), rf.get("frac"), rf.get("peak"), rf.get("unit"),
                " (the chip's issue peak; %s of the kernel's measured mix ceiling), %s lane-instructions per cell" % (
                    (rf.get("mix_ceiling") or {}).get("frac"), rf.get("lane_insts_per_cell"))
                if rf.get("bound") == "valu" else "", (rf.get("hbm") or rf).get("frac"))
    if anchor:
        out["c1_anchor"] = ("error: " + anchor["error"]) if "error" in anchor else (
            "oracle %s reads/s vs reference %s reads/s (1 thread each, %s); GPU vs reference: %s mismatching of 1000 reads; GPU vs oracle identical: %s" % (
                anchor.get("oracle_reads_per_s_1_thread"), anchor.get("reference_reads_per_s_1_thread"), "ecoli_like 4 641 652 nt",
                anchor.get("gpu_vs_reference_mismatching_reads", anchor.get("reference_error", "reference not on this box")),
                anchor.get("gpu_vs_oracle_identical")))
    if isinstance(boundary, dict) and "error" not in boundary:
        g = boundary.get("graph") or {}
        if g.get("reads_per_s") is not None:
            out["dropin_graph_reads_per_s"] = g.get("reads_per_s")
            out["dropin_graph_threads"] = g.get("threads")
    out["config"] = {
        "workload": head["workload"] + " vs GRCh38-like synthetic genome (%d contigs, %d nt%s)" % (
            len(E.lens), E.F, "" if args.no_repeats else ", planted repeats"),
        "value_is": "workload '%s' (%s), %s" % (
            head["name"], head["baseline_config"],
            ( "%d batch(es) in flight per GPU, %s (the one leg --inflight / --host-io asked for)" % (head.get("batches_in_flight", 1), head.get("io", ""))
              if args.inflight > 0 else "one batch at a time, device resident (no overlapped leg ran)" )
            if key is None else
            "%d batches in flight per GPU, %s; the same batches with reads and results resident in HBM: value_150bp_device_resident" % (
                top["batches_in_flight"], top.get("io", ""))),
        "reads_per_s_total": head["reads_per_s_total"], "index_build_s": round(E.t_index, 2),
        "parallelism": "reads partitioned over %d GPU(s) (%s scaling), index replicated, no collective; %s batch(es) in flight per GPU" % (
            E.world, args.scaling, top.get("batches_in_flight", 1)),
        "detail_file": os.path.relpath(dpath, ROOT) if os.path.isabs(dpath) else dpath,
        "kernel_source_hash": kernel_source_hash(),
        "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
    }
    # The top-level roofline is the SINGLE-STREAM leg's (one batch at a time: a launch's duration is the kernel's own, and
    # avg_launch_ms <= that leg's ms_per_step); under overlap a kernel's launch time includes the share of the chip the other
    # batches' kernels took, which says nothing about the kernel -- that leg's figures are in the detail file.
    rf = dict(head["roofline"])
    rf["leg"] = "one batch at a time, device resident: value_%s_single_stream, %s ms per step" % (
        {"150bp": "150bp", "10kb": "10kb", "50kb": "50kb", "illumina": "150bp_illumina"}.get(head["name"], head["name"]), head["ms_per_step"])
    rf.pop("kernel_ms_per_step", None)
    hb = rf.get("hbm")
    if isinstance(hb, dict):
        hb.pop("pmc_replay_refused", None)
    out["roofline"] = rf
    cb = dict(head.get("cpu_baseline") or {})
    cb.pop("port", None)
    if "sample" in cb:
        cb["sample"] = cb["sample"][:160]
    pc = cb.get("parity_check")
    if isinstance(pc, dict):
        pc.pop("what", None)
    out["cpu_baseline"] = cb or None
    return json.dumps(out)


if __name__ == "__main__":
    main()
