#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/b_tests.txt
python bench.py --workload 10kb --boundary-reads 0 --cpu-sample 4 --overlap 0 --steps 3 > gpurun_out/b_10kb.json 2> gpurun_out/b_10kb.err
MA_SEED_LONG_JUMP=0 python bench.py --workload 10kb --boundary-reads 0 --cpu-sample 0 --overlap 0 --steps 3 > gpurun_out/b_10kb_nojump.json 2>/dev/null
python3 tools/ksw_prof.py --workload 10kb --boundary-reads 0 --overlap 0 2>&1 | grep -v "^{" | grep -v amdgpu.ids > gpurun_out/c_prof_jump.txt
