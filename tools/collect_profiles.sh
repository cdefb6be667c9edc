#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box for ONE workload of bench.py: a kernel trace and separate PMC
# passes (HBM bytes, VALU / scalar instructions) as MI355X_MICROARCH.md prescribes (counters in their own runs,
# --kernel-trace only).   usage: bash tools/collect_profiles.sh <tag> <workload: 150bp|10kb|50kb|10kb_pacbio|50kb_nanopore> [preset]
#   -> gpurun_out/prof_<tag>_<workload>/{kernel_stats.csv,summary.txt,pmc_counters.csv}, entry added to profiles-style
#      gpurun_out/prof_<tag>_pmc_traffic.json
set -u
TAG=${1:-r05}
WL=${2:-150bp}
PRESET=${3:-default}
OUT=gpurun_out/prof_${TAG}_$WL
if [ "$PRESET" != default ]; then OUT=${OUT}_$PRESET; fi
mkdir -p $OUT
export TMPDIR=/tmp
case $WL in 150bp) RL=150; RPS=1000000;; 10kb) RL=10000; RPS=200000;; 50kb) RL=50000; RPS=20000;;
  10kb_pacbio) RL=10000; RPS=100000; PRESET=pacbio; OUT=gpurun_out/prof_${TAG}_$WL; mkdir -p $OUT;;
  50kb_nanopore) RL=50000; RPS=10000; PRESET=nanopore; OUT=gpurun_out/prof_${TAG}_$WL; mkdir -p $OUT;; esac
ARGS="bench.py --workload $WL --preset $PRESET --steps 3 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $ARGS > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o pmc --output-format csv -- python3 $ARGS > $OUT/pmc_$C.log 2>&1
done
python3 tools/pmc_summarize.py $OUT 4 gpurun_out/prof_${TAG}_pmc_traffic.json $RL $RPS $PRESET > $OUT/summary.txt 2>&1 || echo "pmc_summarize FAILED for $WL (see $OUT/summary.txt)"
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU $OUT/pmc_SQ_INSTS_SALU
tail -40 $OUT/summary.txt
