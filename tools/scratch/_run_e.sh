#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/e_tests.txt
rm -f gpurun_out/e_exp.txt
for WL in 150bp 50kb 10kb; do
  r=$(python bench.py --workload $WL --steps 5 --warmup 1 --cpu-sample 8 --boundary-reads 0 --overlap 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=j['config']['workloads'][0]; print(j['ms_per_step'], w['roofline']['kernel_ms_per_step'], w['cpu_baseline']['parity_check']['mismatching_reads'], w['roofline'].get('dp_band_cells_per_read_executed'))")
  echo "$WL ms_per_step: $r" >> gpurun_out/e_exp.txt
done
r=$(python bench.py --workload 150bp --preset illumina --steps 5 --warmup 1 --cpu-sample 8 --boundary-reads 0 --overlap 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=j['config']['workloads'][0]; print(j['ms_per_step'], w['roofline']['kernel_ms_per_step'], w['cpu_baseline']['parity_check']['mismatching_reads'], w['roofline'].get('dp_band_cells_per_read_executed'))")
echo "illumina ms_per_step: $r" >> gpurun_out/e_exp.txt
