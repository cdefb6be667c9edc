#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/i_host.txt
: > $O
echo "nproc $(nproc)" >> $O
cat /sys/fs/cgroup/cpu.max >> $O 2>&1
stat() { echo "--- $1" >> $O; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat >> $O 2>&1; cat /proc/loadavg >> $O; grep -E "Dirty|Writeback:|MemFree|Cached:" /proc/meminfo >> $O; }
run() { python bench.py --workload 150bp --steps 2 --warmup 1 --cpu-sample 2 --overlap 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['config']['boundary']
print('$1', 'graph', b['graph']['reads_per_s'], 'sam', b['sam']['reads_per_s'], 'flat2', b['batch_aligner_flat']['inflight_2']['reads_per_s'])" >> $O; }
stat start
run first; stat after1
run second; stat after2
sleep 60; stat idle60
run third; stat after3
