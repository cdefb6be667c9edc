export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3_dp_tests.log
for wl in 50kb 10kb; do
rocprofv3 --kernel-trace -d gpurun_out/tr_$wl -o tr --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/tr_$wl.log 2>&1
python3 tools/launch_list.py gpurun_out/tr_$wl k_chain k_sort_seeds k_soc k_stitch > gpurun_out/r03b_launch_timeline_$wl.txt; rm -rf gpurun_out/tr_$wl
done
