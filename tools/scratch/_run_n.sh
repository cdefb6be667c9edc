#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/n_top.txt
: > $O
( python bench.py --workload 150bp --steps 2 --warmup 1 --cpu-sample 2 --overlap 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['config']['boundary']
print('graph', b['graph']['reads_per_s'], b['graph'].get('device_side'))" > gpurun_out/n_res.txt ) &
BP=$!
for i in $(seq 1 60); do
  sleep 4
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "=== t=$((i*4))s $(cat /proc/loadavg)" >> $O
  top -H -b -n 1 -w 160 2>/dev/null | sed -n 7,22p >> $O
done
wait $BP
cat gpurun_out/n_res.txt >> $O
