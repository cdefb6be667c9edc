#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/d_slow_batch.txt
run() { python bench.py --workload $1 --boundary-reads 0 --cpu-sample $2 --overlap 0 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['config']['workloads'][0]
print('$1 $3','k_seed',w['roofline']['kernel_ms_per_step']['k_seed'],'step',w['ms_per_step'], (w.get('cpu_baseline') or {}).get('parity_check',{}).get('mismatching_reads'))" >> gpurun_out/d_slow_batch.txt; }
MA_SEED_TASKS=1 run 10kb 0 "tasks"
MA_SEED_TASKS=1 MA_SEED_SLOW_BATCH=16 run 10kb 0 "tasks sb16"
MA_SEED_TASKS=1 MA_SEED_SLOW_BATCH=32 run 10kb 0 "tasks sb32"
rocprofv3 --kernel-trace -d gpurun_out/tr_t -o tr --output-format csv -- python3 bench.py --workload 50kb --steps 1 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > /dev/null 2>&1
python3 tools/launch_list.py gpurun_out/tr_t k_seed_tasks k_task > gpurun_out/d_tasks_timeline_50kb.txt; rm -rf gpurun_out/tr_t
