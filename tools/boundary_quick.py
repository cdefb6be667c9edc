#!/usr/bin/env python3
"""Runs examples/ma_boundary_bench on a scaled-down genome (quick iteration on the host-side boundary code).
usage: python tools/boundary_quick.py [genome scale=0.1] [reads=1000000] [graph threads]"""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GRCH38
import torch, ma_amd
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
reads = sys.argv[2] if len(sys.argv) > 2 else "1000000"
L = ma_amd.lib()
lens = np.array([max(1000, int(x * scale)) for x in GRCH38], dtype=np.uint64)
F = int(lens.sum())
g = torch.empty(F, dtype=torch.uint8, device="cuda")
assert L.ma_synth_genome_device(C.c_uint64(2), C.c_uint64(F), C.c_int32(1), C.c_void_p(g.data_ptr())) == 0
idx = ma_amd.Index.build_device(lens, g.data_ptr())
d = tempfile.mkdtemp(prefix="ma_bq_")
idx.store(os.path.join(d, "idx"))
idx.close()
del g
torch.cuda.empty_cache()
cmd = [os.path.join(ROOT, "examples", "ma_boundary_bench"), os.path.join(d, "idx"), reads, "150", "default", "0"] + sys.argv[3:4]
print(subprocess.run(cmd, capture_output=True, text=True))
