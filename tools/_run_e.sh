#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/e_tests.txt
rm -f gpurun_out/e_exp.txt
for WL in 50kb 10kb 150bp; do
  r=$(python bench.py --workload $WL --steps 3 --warmup 1 --cpu-sample 4 --boundary-reads 0 --overlap 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=j['config']['workloads'][0]; print(j['ms_per_step'], w['roofline']['kernel_ms_per_step'], w['cpu_baseline']['parity_check']['mismatching_reads'])")
  echo "$WL ms_per_step: $r" >> gpurun_out/e_exp.txt
done
bash tools/_run_f.sh
