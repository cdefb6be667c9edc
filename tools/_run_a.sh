#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/a_tests.txt
python -m pytest tests -m gpu -x -q -k "smem or illumina or SMEM or technique" 2>&1 | tail -5 >> gpurun_out/a_tests.txt
python bench.py --workload 150bp --preset illumina --boundary-reads 0 --cpu-sample 8 > gpurun_out/a_illumina.json 2> gpurun_out/a_illumina.err
MA_SMEM_MERGE=0 python bench.py --workload 150bp --preset illumina --boundary-reads 0 --cpu-sample 0 --overlap 0 > gpurun_out/a_illumina_nomerge.json 2>/dev/null
