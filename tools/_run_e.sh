#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/e_tests.txt
for v in "" noneed; do
  lib=""; [ -n "$v" ] && lib=$PWD/tools/_prof/libma_amd_$v.so
  for WL in 10kb 50kb; do
  r=$(MA_AMD_LIB=$lib python bench.py --workload $WL --steps 2 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['kernel_ms_per_step']['k_ksw'])")
  echo "variant=${v:-product} $WL ms_per_step k_ksw_ms: $r" >> gpurun_out/e_exp.txt
  done
done
