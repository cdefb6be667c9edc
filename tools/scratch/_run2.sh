export TMPDIR=/tmp
python -m pytest tests/test_host_graph.py tests/test_sam_writer.py tests/test_f4_host.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_host_tests.log
python bench.py --workload 150bp --steps 5 --warmup 1 --cpu-sample 4000 --cpu-threads-sweep 0 > gpurun_out/r3_b150.json 2> gpurun_out/r3_b150.err
