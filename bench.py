#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the MI355X seed-and-extend path on BASELINE.json's workload.

A "step" is one pass of the whole hot path (FMD seeding -> seed extraction -> SoC/harmonization ->
banded-DP gap fill/extension -> mapping quality) over one batch of synthetic reads that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: 150 bp Illumina-like reads (0.5 %
substitutions) against a GRCh38-sized synthetic genome (24 contigs, 3.09 Gnt, planted repeat
families), Default preset, 10 steps x 1 M reads = 10 M reads on one MI355X.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL, used only for the barrier and the
max-over-ranks time); reads are partitioned by index, the index is replicated, there is no data-path
collective.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616,
          64444167, 46709983, 50818468, 156040895, 57227415]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
GATHER_CEILING_GBLOCKS = 51.8  # measured on MI355X by tools/gups.hip: random 64-B blocks/s, all CUs (profiles/r01_gups.txt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-step", type=int, default=0, help="default: 1 M reads (<= 1 kb), else 2 Gbase worth of reads")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub", type=float, default=0.005)
    ap.add_argument("--ins", type=float, default=0.0)
    ap.add_argument("--dele", type=float, default=0.0)
    ap.add_argument("--genome-scale", type=float, default=1.0, help="fraction of GRCh38 contig lengths (tests only)")
    ap.add_argument("--preset", default="default")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1 auto, 0 off)")
    ap.add_argument("--no-repeats", action="store_true")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU (own stream + host thread each): while one batch is in its "
                         "VALU-bound DP kernels another runs its memory-bound seeding / chaining")
    args = ap.parse_args()

    import torch
    import ma_amd
    from ma_amd.shard import weak_shard_first_index, reduce_timing_and_counts

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # test hook: MA_BENCH_ONE_DEVICE=1 runs all ranks on GPU 0 over gloo, so that the multi-rank code path (barriers,
    # MAX/SUM reductions, rank-0 reporting) can be exercised on a one-GPU box; the driver's runs use RCCL, one GPU per rank
    one_dev = os.environ.get("MA_BENCH_ONE_DEVICE") == "1"
    gpu = 0 if one_dev else local_rank
    if world > 1:
        import torch.distributed as dist
        if one_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", gpu))
    torch.cuda.set_device(gpu)
    ma_amd.set_device(gpu)
    dev = torch.device("cuda", gpu)
    local_rank = gpu
    L = ma_amd.lib()

    def chk(rc):
        if rc != 0:
            raise RuntimeError(L.ma_last_error().decode())

    # ---- synthetic genome + index (not timed: one-off preprocessing) -----------------
    lens = np.array([max(1000, int(x * args.genome_scale)) for x in GRCH38], dtype=np.uint64)
    F = int(lens.sum())
    t0 = time.perf_counter()
    g = torch.empty(F, dtype=torch.uint8, device=dev)
    chk(L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(F), C.c_int32(0 if args.no_repeats else 1),
                                 C.c_void_p(g.data_ptr())))
    idx = ma_amd.Index.build_device(lens, g.data_ptr())
    del g
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_index = time.perf_counter() - t0

    # ---- reads: every rank owns steps x reads_per_step reads (weak scaling), resident in HBM ---------
    B = args.reads_per_step if args.reads_per_step > 0 else (1000000 if args.read_len <= 1000 else max(10000, int(2e9 / args.read_len)))
    K, W = args.steps, args.warmup
    n_reads = B * K
    cap = int(n_reads * (args.read_len * (1.0 + 2 * args.ins) + 8)) + 1024
    codes = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = torch.empty(n_reads + 1, dtype=torch.int64, device=dev)
    nb = C.c_uint64()
    seed = 11 if args.read_len <= 1000 else (12 if args.read_len <= 20000 else 13)
    chk(L.ma_synth_reads_device(idx.h, C.c_uint64(seed), C.c_uint64(n_reads), C.c_uint32(args.read_len),
                                C.c_double(args.sub), C.c_double(args.ins), C.c_double(args.dele),
                                C.c_uint64(weak_shard_first_index(n_reads, rank)), C.c_void_p(codes.data_ptr()), C.c_void_p(offs.data_ptr()),
                                C.c_uint64(cap), C.byref(nb)))
    offs_h = offs.cpu().numpy().astype(np.uint64)

    P = ma_amd.Params.preset(args.preset)
    max_bases = int((offs_h[B::B] - offs_h[:-1:B]).max()) if K > 0 else 0
    NB = max(1, min(args.inflight, K))
    batches = []
    for i in range(NB):
        bt = ma_amd.Batch(idx, P, B, max_bases + 64)
        st = torch.cuda.current_stream() if NB == 1 else torch.cuda.Stream()
        bt.set_stream(st.cuda_stream)
        bt.enable_timing(True)
        batches.append((bt, st))

    def step(i, k):
        bt = batches[i][0]
        lo = k * B
        nbases = int(offs_h[lo + B] - offs_h[lo])
        bt.set_reads_device(codes.data_ptr(), offs.data_ptr() + 8 * lo, B, nbases)
        bt.align()
        bt.sync()

    # warm-up: W steps on every batch object (each sizes its own buffers); untimed
    for w in range(W):
        for i in range(NB):
            step(i, (w + i) % K)

    import threading
    acc = [dict(kms=np.zeros(8), ctr=np.zeros(8), segs=0, aligned=0, err=None) for _ in range(NB)]

    def worker(i):
        try:
            torch.cuda.set_device(local_rank)
            a = acc[i]
            for k in range(i, K, NB):
                tk = time.perf_counter()
                step(i, k)
                bt = batches[i][0]
                km = bt.kernel_ms().astype(np.float64)
                if os.environ.get("MA_BENCH_VERBOSE"):
                    print("step %d wall %.1f ms, stage ms %s" % (k, (time.perf_counter() - tk) * 1e3, np.round(km[:6], 2).tolist()),
                          file=sys.stderr, flush=True)
                a["kms"] += km
                a["ctr"] += bt.counters().astype(np.float64)
                c = bt.counts()
                a["aligned"] += c["aligned_reads"]
                a["segs"] += c["segments"]
        except Exception as e:  # surfaced after the join
            acc[i]["err"] = e

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
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
    for a in acc:
        if a["err"] is not None:
            raise a["err"]
    kms = sum(a["kms"] for a in acc)
    ctr = sum(a["ctr"] for a in acc)
    segs = sum(a["segs"] for a in acc)
    aligned = sum(a["aligned"] for a in acc)
    dt, (aligned_all,) = reduce_timing_and_counts(dist, torch.device("cpu") if one_dev else dev, dt, [aligned])

    # ---- roofline of the dominant kernel (HIP events on the launch stream, averaged over the K launches) --
    names = ["k_seed", "k_seed_rows+k_lf_walk+k_seed_final", "k_chain", "k_dp_enum", "k_ksw", "k_stitch+k_finish"]
    total_bases = float(offs_h[n_reads] - offs_h[0])
    alg = [64.0 * ctr[1] + total_bases + 40.0 * segs,  # seeding: occ blocks + read bases + segments out
           64.0 * ctr[2] + 8.0 * ctr[3] + 48.0 * ctr[3],  # SA lookup: LF blocks + SA sample + seed out
           0.0, 0.0,
           ctr[6] + ctr[4] + ctr[7],  # DP: sequences in + 1 B per band cell + back-trace reads + cigar out
           0.0]
    dom = int(np.argmax(kms[:6]))
    avg_s = kms[dom] / 1e3 / max(K, 1)
    ach = (alg[dom] / max(K, 1)) / avg_s / 1e9 if avg_s > 0 else 0.0
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see
    # DESIGN.md 3.4); only meaningful for the workload those passes were taken on
    traffic = None
    valu_issue = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pt = json.load(f)
        if pt.get("workload_key") == [args.read_len, B, args.preset, args.genome_scale]:
            traffic = pt["bytes_per_launch"].get(names[dom])
            vi = pt.get("valu_wave_insts_per_launch", {}).get(names[dom])
            if vi and avg_s > 0:
                # the DP kernels are VALU-issue bound, not HBM bound: instructions from the committed PMC pass over
                # the live launch time, against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
                valu_issue = {"wave_insts_per_launch": vi, "achieved_Ginst_s": round(vi / avg_s / 1e9, 1),
                              "peak_Ginst_s": 614.4, "frac": round(vi / avg_s / 1e9 / 614.4, 3)}
    except (OSError, ValueError, KeyError):
        traffic = None
    roofline = {"kernel": names[dom], "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "valu_issue": valu_issue,
                "avg_launch_ms": round(kms[dom] / max(K, 1), 3),
                "algorithmic_bytes_per_launch": int(alg[dom] / max(K, 1)),
                "kernel_ms_per_step": {names[i]: round(kms[i] / max(K, 1), 3) for i in range(6)},
                "seeding_GBps": round((alg[0] / max(K, 1)) / (kms[0] / 1e3 / max(K, 1)) / 1e9, 2) if kms[0] > 0 else 0.0,
                "seeding_frac_of_gather_ceiling": round(ctr[1] / (kms[0] / 1e3) / 1e9 / GATHER_CEILING_GBLOCKS, 4) if kms[0] > 0 else 0.0,
                "dp_GCUPS": round(ctr[4] / (kms[4] / 1e3) / 1e9, 3) if kms[4] > 0 else 0.0}

    # ---- CPU baseline: the oracle (bit-exact restatement of the reference) on this host's cores -----------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from ma_testlib import OrIndex, or_params
        ncores = os.cpu_count() or 1
        S = args.cpu_sample if args.cpu_sample > 0 else min(n_reads, max(2000, int(4000 * ncores * 150 / max(args.read_len, 1))))
        oidx = OrIndex.from_parts(idx.download())
        hb = int(offs_h[S])
        rc = codes[:hb].cpu().numpy()
        reads = [rc[int(offs_h[i]):int(offs_h[i + 1])] for i in range(S)]
        op = or_params(args.preset, 1)
        t1 = time.perf_counter()
        res = oidx.align(reads, op, threads=ncores)
        tc = time.perf_counter() - t1
        cpu = {"value": round(res["n_aligned"] / tc, 1), "unit": "aligned reads/s", "cores": ncores, "kind": "port",
               "sample": "first %d reads of the same workload, oracle with %d threads, %.1f s" % (S, ncores, tc),
               # the reference computes every diagonal of an extension; the GPU path stops once ez.max is final
               "dp_band_cells_per_read": round(float(res["counters"][4]) / S, 1)}
        roofline["dp_band_cells_per_read_executed"] = round(ctr[4] / max(n_reads, 1), 1)
        # ---- the REAL reference (compiled from its own sources into oracle/_ref by oracle/Makefile.ref) on the same
        # sample: the GPU-built index is written in the reference's file formats, the reference's own loaders read it,
        # its modules (BinarySeeding .. MappingQuality) run on all host threads (oracle/ref_dump.cpp timeidx)
        ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
        if os.path.exists(ref_dump) and not os.environ.get("MA_BENCH_NO_REFERENCE"):
            import shutil
            import subprocess
            import tempfile
            from ma_testlib import write_case
            td = tempfile.mkdtemp(prefix="ma_ref_")
            try:
                t2 = time.perf_counter()
                idx.store(os.path.join(td, "idx"))
                write_case(os.path.join(td, "reads.case"), [], reads)
                out = subprocess.run([ref_dump, "timeidx", os.path.join(td, "idx"), os.path.join(td, "reads.case"), args.preset,
                                      str(ncores)], capture_output=True, text=True, timeout=1800)
                m = __import__("re").search(r"(\d+) reads \((\d+) aligned\) in ([0-9.]+) s on (\d+) threads", out.stdout)
                if out.returncode == 0 and m:
                    tr = float(m.group(3))
                    cpu = dict(cpu, value=round(int(m.group(2)) / tr, 1), kind="reference",
                               sample="first %d reads of the same workload, the reference's own modules on %d threads, %.1f s "
                                      "(index written by the GPU builder and loaded by the reference's loaders: %.0f s, not "
                                      "counted)" % (S, ncores, tr, time.perf_counter() - t2 - tr),
                               port={"value": cpu["value"], "sample": cpu["sample"]})
                else:
                    cpu["reference_error"] = (out.stderr or out.stdout)[-300:]
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
        cpu["parity_check"] = {"reads": Pn, "alignments": na, "alignment_ops": nops, "mapq_records": nm,
                               "mismatching_reads": bad,
                               "what": "GPU vs oracle on the first reads of step 0: every NeedlemanWunsch alignment (positions, score, "
                                       "ops) and MappingQuality record (flags, mapq bits)"}

    if rank == 0:
        out = {
            "metric": "aligned reads/sec (whole node), 150bp & 10kb synthetic vs GRCh38",
            "value": round(aligned_all / dt, 1), "unit": "aligned reads/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / max(K, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int8 (2-bit BWT ranks, int8 DP differences, int16/32 scores)",
            "data": "synthetic",
            "config": {"workload": "%d x %d bp reads (%.2f%% sub, %.2f%% ins, %.2f%% del) vs GRCh38-like synthetic genome "
                       "(%d contigs, %d nt%s), %s preset, %d reads/step per GPU" % (
                           n_reads * world, args.read_len, 100 * args.sub, 100 * args.ins, 100 * args.dele, len(lens), F,
                           "" if args.no_repeats else ", planted repeats", args.preset, B),
                       "reads_per_s_total": round(n_reads * world / dt, 1), "index_build_s": round(t_index, 2),
                       "parallelism": "reads partitioned over %d GPU(s), index replicated, no collective; %d batches in "
                                      "flight per GPU" % (world, NB)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
