#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/tr_t -o tr --output-format csv -- python3 bench.py --workload 50kb --steps 1 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > /dev/null 2>&1
python3 tools/launch_list.py gpurun_out/tr_t k_ksw k_job_cost > gpurun_out/f_timeline_50kb.txt; rm -rf gpurun_out/tr_t
