#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/d_lanes.txt
run() { python bench.py --workload $1 --boundary-reads 0 --cpu-sample 0 --overlap 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['config']['workloads'][0]
print('$1 $2 k_seed',w['roofline']['kernel_ms_per_step']['k_seed'],'step',w['ms_per_step'])" >> gpurun_out/d_lanes.txt; }
for L in 32768 65536 98304 131072 163840; do MA_SEED_LANES=$L run 10kb "lanes $L"; done
for L in 131072 262144 393216; do MA_SEED_LANES=$L run 150bp "lanes $L"; done
