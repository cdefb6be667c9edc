#!/bin/bash
# Calibration evidence of the roofline, one GPU call:
#  - tools/valu_mix: measured issue rate of the DP kernels' VALU opcodes and of their per-diagonal mix
#  - tools/gups: random-gather ceiling (32 / 64 / 128 B per block) and, under rocprofv3 --pmc, raw FETCH_SIZE /
#    WRITE_SIZE against a KNOWN byte count for exactly the occ-block access pattern (4 x dwordx4 per lane)
# usage: bash tools/calibrate.sh <tag>  -> gpurun_out/calib_<tag>/
set -u
TAG=${1:-r02}
OUT=gpurun_out/calib_$TAG
mkdir -p $OUT tools/_prof
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/valu_mix.hip -o tools/_prof/valu_mix 2>/dev/null
hipcc --offload-arch=gfx950 -O3 tools/gups.hip -o tools/_prof/gups 2>/dev/null
tools/_prof/valu_mix > $OUT/valu_mix.txt 2>&1
tools/_prof/gups > $OUT/gups.txt 2>&1
for cfg in "stream" "one 64 0 8 1000" "one 64 1 8 1000" "one 32 0 8 1000" "one 128 0 8 1000"; do
  name=$(echo $cfg | tr ' ' '_')
  for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
    rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${name}_$C -o pmc --output-format csv -- tools/_prof/gups $cfg > $OUT/pmc_${name}_$C.log 2>&1
  done
done
python3 tools/calibrate_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
