#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box: a kernel trace of the default bench and separate PMC passes
# (HBM bytes, VALU instructions) as MI355X_MICROARCH.md prescribes (counters in their own runs, --kernel-trace only).
# usage: bash tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $ARGS > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc --output-format csv -- python3 $ARGS > $OUT/pmc_$C.log 2>&1
done
python3 tools/pmc_summarize.py $OUT 4 $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1
cp $OUT/trace/trace_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv ;
tail -30 $OUT/summary.txt
