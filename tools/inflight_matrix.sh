for cfg in "1 32" "1 12" "2 12" "2 8" "3 12" "2 16" "3 8"; do set -- $cfg
  MA_KSW_WAVES_PER_CU=$2 python bench.py --steps 12 --warmup 2 --cpu-sample 0 --inflight $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('inflight $1 perCu $2', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
done
