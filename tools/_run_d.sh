#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/d_smem.txt
for SB in 1 2 4 8 16 32; do
MA_SEED_SLOW_BATCH=$SB python bench.py --workload 150bp --preset illumina --boundary-reads 0 --cpu-sample 0 --overlap 0 --steps 4 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['config']['workloads'][0]
print('slow_batch $SB k_seed',w['roofline']['kernel_ms_per_step']['k_seed'],'step',w['ms_per_step'])" >> gpurun_out/d_smem.txt
done
