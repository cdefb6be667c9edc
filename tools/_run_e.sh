#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/e_tests.txt
rm -f gpurun_out/e_exp.txt
for WL in 50kb 10kb; do
  r=$(python bench.py --workload $WL --steps 3 --warmup 1 --cpu-sample 4 --boundary-reads 0 --overlap 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=j['config']['workloads'][0]; print(j['ms_per_step'], w['roofline']['kernel_ms_per_step'], w['cpu_baseline']['parity_check']['mismatching_reads'])")
  echo "$WL ms_per_step: $r" >> gpurun_out/e_exp.txt
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/tr_t -o tr --output-format csv -- python3 bench.py --workload 50kb --steps 1 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > /dev/null 2>&1
python3 tools/launch_list.py gpurun_out/tr_t k_soc k_chain k_sort_seeds > gpurun_out/f_timeline_50kb.txt; rm -rf gpurun_out/tr_t
