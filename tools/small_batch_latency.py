#!/usr/bin/env python3
"""Wall time of every stage call for small device batches (what the per-read graph funnel produces): where do the
milliseconds of a 600-read batch go?   usage: python tools/small_batch_latency.py [genome scale=0.1]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GRCH38
import torch, ma_amd
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
L = ma_amd.lib()
lens = np.array([max(1000, int(x * scale)) for x in GRCH38], dtype=np.uint64)
F = int(lens.sum())
g = torch.empty(F, dtype=torch.uint8, device="cuda")
assert L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(F), C.c_int32(1), C.c_void_p(g.data_ptr())) == 0
idx = ma_amd.Index.build_device(lens, g.data_ptr())
del g
N = 8192
codes = torch.empty(N * 160, dtype=torch.uint8, device="cuda")
offs = torch.empty(N + 1, dtype=torch.int64, device="cuda")
nb = C.c_uint64()
assert L.ma_synth_reads_device(idx.h, C.c_uint64(11), C.c_uint64(N), C.c_uint32(150), C.c_double(0.005), C.c_double(0), C.c_double(0),
                               C.c_uint64(0), C.c_void_p(codes.data_ptr()), C.c_void_p(offs.data_ptr()), C.c_uint64(N * 160), C.byref(nb)) == 0
hc = codes.cpu().numpy(); ho = offs.cpu().numpy()
P = ma_amd.Params.preset("default")
for blocking in (0, 1):
    for n in (64, 600, 4096):
        reads = [hc[int(ho[i]):int(ho[i + 1])] for i in range(n)]
        b = ma_amd.Batch(idx, P, n, n * 160)
        L.ma_batch_set_blocking_sync(b.h, C.c_int(blocking))
        b.enable_timing(True)
        acc = {}
        for it in range(25):
            t = [time.perf_counter()]
            b.set_reads(reads); t.append(time.perf_counter())
            b.seed(); t.append(time.perf_counter())
            b.extract(); t.append(time.perf_counter())
            b.chain(); t.append(time.perf_counter())
            b.dp(); t.append(time.perf_counter())
            b.sync(); t.append(time.perf_counter())
            b.mapq_alignments(); t.append(time.perf_counter())
            if it >= 5:
                for k, name in enumerate(("set_reads", "seed", "extract", "chain", "dp", "sync", "download")):
                    acc[name] = acc.get(name, 0.0) + (t[k + 1] - t[k]) * 1e3 / 20
                acc["kernel_events"] = acc.get("kernel_events", 0) + float(b.kernel_ms()[:6].sum()) / 20
        print("blocking=%d n=%d  total %.2f ms: %s" % (blocking, n, sum(v for k, v in acc.items() if k != "kernel_events"),
              " ".join("%s %.2f" % (k, v) for k, v in acc.items())))
        b.close()
