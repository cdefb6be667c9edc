#!/bin/bash
mkdir -p gpurun_out
python bench.py > gpurun_out/h_bench_default.json 2> gpurun_out/h_bench_default.err
