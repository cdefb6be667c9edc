#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r03_gpu_tests.txt
python bench.py > gpurun_out/h_bench_default.json 2> gpurun_out/h_bench_default.err
