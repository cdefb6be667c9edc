#!/bin/bash
# Everything the round's numbers come from, in one GPU call: parity tests, rocprofv3 evidence, the three bench lines.
TAG=${1:-r01}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.txt
cat gpurun_out/${TAG}_gpu_tests.txt
bash tools/collect_profiles.sh $TAG > /dev/null 2>&1
python bench.py 2> gpurun_out/${TAG}_bench_default.err | tail -1 > gpurun_out/${TAG}_bench_default.json
python bench.py --read-len 10000 --sub 0.004 --ins 0.003 --dele 0.003 --steps 5 --warmup 1 2> gpurun_out/${TAG}_bench_10kb.err | tail -1 > gpurun_out/${TAG}_bench_10kb.json
python bench.py --preset illumina --steps 5 --warmup 1 2> gpurun_out/${TAG}_bench_illumina.err | tail -1 > gpurun_out/${TAG}_bench_illumina.json
for f in default 10kb illumina; do python3 -c "
import json,sys
j=json.load(open('gpurun_out/${TAG}_bench_$f.json')); c=j.get('cpu_baseline') or {}
print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'], (j['roofline'].get('valu_issue') or {}).get('frac'), c.get('value'), (c.get('parity_check') or {}).get('mismatching_reads'), (c.get('parity_check') or {}).get('reads'))"; done
