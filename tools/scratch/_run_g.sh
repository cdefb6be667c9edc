#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/g_ab.txt
run() { python bench.py --workload 150bp --steps 2 --warmup 1 --cpu-sample 2 --overlap 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['config']['boundary']
print('$1', 'graph', b['graph']['reads_per_s'], 'sam', b['sam']['reads_per_s'], 'flat2', b['batch_aligner_flat']['inflight_2']['reads_per_s'], 'aligner1', b['batch_aligner']['inflight_1']['reads_per_s'], 'step', d['config']['workloads'][0]['ms_per_step'])" >> gpurun_out/g_ab.txt; }
run new
LD_PRELOAD=$PWD/tools/_prof/old/libma_amd.so run old
run new
LD_PRELOAD=$PWD/tools/_prof/old/libma_amd.so run old
