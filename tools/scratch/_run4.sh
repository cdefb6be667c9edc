export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_all_tests.log
python tools/boundary_quick.py 0.05 1000000 > gpurun_out/bq.log 2>&1
