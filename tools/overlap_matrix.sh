#!/bin/bash
# Two or more batches in flight on their own streams overlap the VALU-bound DP kernels of one batch with the memory-bound
# seeding / chaining of another.  Sweeps batches in flight x DP turn-taking (MA_DP_EXCLUSIVE: one DP stage at a time per
# device) on the 150 bp workload.   usage: bash tools/overlap_matrix.sh > gpurun_out/overlap_matrix.txt
for excl in 1 0; do
  for inflight in 1 2 3 4; do
    v=$(MA_DP_EXCLUSIVE=$excl python bench.py --workload 150bp --steps 24 --warmup 2 --cpu-sample 0 --boundary-reads 0 --inflight $inflight 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step']['k_ksw'])")
    echo "dp_exclusive=$excl inflight=$inflight reads_per_s ms_per_step k_ksw_ms: $v"
  done
done
