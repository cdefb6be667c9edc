for t in 1 4 8 12 16 24; do
  MA_SEED_SLOW_BATCH=$t python bench.py --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('slow_batch $t', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step']['k_seed'], j['roofline']['seeding_frac_of_gather_ceiling'])"
done
MA_SEED_SLOW_BATCH=8 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --preset illumina 2>/dev/null | tail -1 | cut -c1-200
MA_SEED_SLOW_BATCH=1 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --preset illumina 2>/dev/null | tail -1 | cut -c1-200
