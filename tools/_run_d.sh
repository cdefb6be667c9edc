#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/d_tests.txt
: > gpurun_out/d_slow_batch.txt
for SB in 4 8 16 32; do
  MA_SEED_SLOW_BATCH=$SB python bench.py --workload 50kb --boundary-reads 0 --cpu-sample 0 --overlap 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['config']['workloads'][0]
print('50kb slow_batch',$SB,'k_seed',w['roofline']['kernel_ms_per_step']['k_seed'],'step',w['ms_per_step'])" >> gpurun_out/d_slow_batch.txt
done
MA_SEED_WINDOW=0 python bench.py --workload 50kb --boundary-reads 0 --cpu-sample 0 --overlap 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['config']['workloads'][0]
print('50kb no window','k_seed',w['roofline']['kernel_ms_per_step']['k_seed'],'step',w['ms_per_step'])" >> gpurun_out/d_slow_batch.txt
python bench.py --workload 10kb --boundary-reads 0 --cpu-sample 4 --overlap 0 --steps 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['config']['workloads'][0]
print('10kb','k_seed',w['roofline']['kernel_ms_per_step']['k_seed'],'step',w['ms_per_step'], w['cpu_baseline']['parity_check']['mismatching_reads'])" >> gpurun_out/d_slow_batch.txt
