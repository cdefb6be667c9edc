export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3_dp_tests.log
for wl in 50kb 10kb 150bp; do
  python bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/ab_${wl}_conc.json 2> gpurun_out/ab_${wl}_conc.err
done
