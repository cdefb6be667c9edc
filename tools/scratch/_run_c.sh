#!/bin/bash
mkdir -p gpurun_out
python3 tools/ksw_prof.py --workload 10kb --boundary-reads 0 --overlap 0 2>&1 | grep -v "^{" | grep -v amdgpu.ids > gpurun_out/c_prof_10kb.txt
MA_SEED_TASKS=0 python3 tools/ksw_prof.py --workload 50kb --boundary-reads 0 --overlap 0 2>&1 | grep -v "^{" | grep -v amdgpu.ids > gpurun_out/c_prof_50kb_notasks.txt
