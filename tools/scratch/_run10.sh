export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3_dp_tests.log
python bench.py --workload 50kb --steps 4 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/ab_50kb_conc.json 2> gpurun_out/ab_50kb_conc.err
