export TMPDIR=/tmp
python -m pytest tests/test_host_graph.py tests/test_ref_binding.py tests/test_f4_host.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_host_tests.log
python tools/boundary_quick.py 0.05 1000000 > gpurun_out/bq.log 2>&1
