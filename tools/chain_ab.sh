#!/bin/bash
# A/B of the chain stage's wave kernels on the long-read workloads (one batch at a time, device resident): everything one read per lane
# (MA_CHAIN_WAVE_SORT=0), the sorts by one wavefront per read and the sweep by lanes (MA_SOC_WAVE=0), both by wavefronts (default).
#   usage (GPU box): bash tools/chain_ab.sh > gpurun_out/r06_chain_wave_ab.txt
for wl in 50kb_nanopore 50kb 10kb_pacbio; do
  for cfg in "MA_CHAIN_WAVE_SORT=0" "MA_SOC_WAVE=0" "default=1"; do
    env ${cfg} python bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --boundary-reads 0 2>&1 | grep "^bench detail" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().split('bench detail: ', 1)[1])
w = d['workloads'][0]; k = w['roofline']['kernel_ms_per_step']
print('%-14s %-22s %9.1f reads/s  %7.1f ms/step   chain stage %6.1f ms   k_seed %6.1f ms' % ('$wl', '$cfg', w['value'], w['ms_per_step'], k['k_chain'], k['k_seed']))"
  done
done
