"""N > 1 path on CPU: two gloo ranks each align their shard of the reads (with the oracle standing in
for the GPU engine) and the reduced results must equal the single-process run."""
import os
import subprocess
import sys

import numpy as np

from ma_testlib import ROOT

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from ma_amd.shard import shard_range, reduce_timing_and_counts
from ma_testlib import OrIndex, or_params, rand_genome, sample_reads
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = rand_genome(1, [40000, 30000])
reads = sample_reads(g, 61, 150, 2, sub=0.01)
lo, hi = shard_range(len(reads), world, rank)
res = OrIndex.build(g).align(reads[lo:hi], or_params())
dist.barrier()
dt, (aligned, nal) = reduce_timing_and_counts(dist, torch.device("cpu"), 1.0 + rank, [res["n_aligned"], len(res["alns"])])
scores = [int(x) for x in res["alns"]["score"]]
gathered = [None] * world
dist.all_gather_object(gathered, (lo, hi, scores))
if rank == 0:
    print(json.dumps({"dt": dt, "aligned": aligned, "nal": nal, "parts": gathered}))
dist.destroy_process_group()
'''


def test_two_rank_partition_matches_single_process(tmp_path):
    from ma_amd.shard import shard_range
    assert [shard_range(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert shard_range(5, 8, 7) == (5, 5)
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29531", str(w), ROOT], env=env,
                                  stderr=subprocess.DEVNULL).decode()
    import json
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ma_testlib import OrIndex, or_params, rand_genome, sample_reads
    g = rand_genome(1, [40000, 30000])
    reads = sample_reads(g, 61, 150, 2, sub=0.01)
    res = OrIndex.build(g).align(reads, or_params())
    assert r["dt"] == 2.0  # MAX over ranks
    assert r["aligned"] == res["n_aligned"] and r["nal"] == len(res["alns"])
    parts = sorted(r["parts"])
    assert parts[0][0] == 0 and parts[-1][1] == len(reads) and parts[0][1] == parts[1][0]
    assert parts[0][2] + parts[1][2] == [int(x) for x in res["alns"]["score"]]
