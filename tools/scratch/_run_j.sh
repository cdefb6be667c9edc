#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/j_tokens.txt
: > $O
run() { python bench.py --workload 150bp --steps 2 --warmup 1 --cpu-sample 2 --overlap 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['config']['boundary']
print('$1', 'graph', b['graph']['reads_per_s'], 'batches', b['graph']['device_batches'], 'sam', b['sam']['reads_per_s'], 'flat2', b['batch_aligner_flat']['inflight_2']['reads_per_s'], 'throttled', b.get('cfs_throttled',{}).get('periods'), b.get('cfs_throttled',{}).get('thread_seconds'))" >> $O; }
run tokens_auto
MA_RUN_TOKENS=0 run no_limit
run tokens_auto
MA_RUN_TOKENS=0 run no_limit
MA_RUN_TOKENS=32 run tokens_32
MA_RUN_TOKENS=12 run tokens_12
