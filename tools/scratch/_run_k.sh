#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/k_graph.txt
: > $O
run() { python bench.py --workload 150bp --steps 2 --warmup 1 --cpu-sample 2 --overlap 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['config']['boundary']
print('$1', 'graph', b['graph']['reads_per_s'], 'batches', b['graph']['device_batches'], '|', b['graph'].get('device_side'))" >> $O; }
run r1; run r2; run r3; run r4
