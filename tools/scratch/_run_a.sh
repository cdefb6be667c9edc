#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/a_tests.txt
python bench.py --workload 150bp --preset illumina --boundary-reads 0 --cpu-sample 8 --steps 5 > gpurun_out/a_illumina.json 2> gpurun_out/a_illumina.err
python bench.py --workload 150bp --boundary-reads 0 --cpu-sample 8 --steps 5 --overlap 0 > gpurun_out/a_default.json 2>/dev/null
