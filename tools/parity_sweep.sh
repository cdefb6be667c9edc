# full-size parity at other read lengths / error profiles (bench.py's parity_check: GPU vs oracle, bit for bit)
# (round 5: 250 / 150 / 100 bp also exercise the proven narrow band on reads with indels and with 2-3 % errors)
# (round 6: 2 kb, 5 kb and 20 kb reads at 0.7 - 4 % errors put long extension jobs of other shapes on the band of 120)
for cfg in "250 0.01 0.002 0.002" "150 0.02 0.005 0.005" "1000 0.02 0.005 0.005" "100 0.03 0.0 0.0" "3000 0.05 0.02 0.02" "2000 0.02 0.01 0.01" "5000 0.01 0.005 0.005" "20000 0.003 0.002 0.002"; do set -- $cfg
  MA_BENCH_NO_REFERENCE=1 python bench.py --read-len $1 --sub $2 --ins $3 --dele $4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); c = j['cpu_baseline']
print('len $1 sub $2 ins $3 del $4:', round(j['value']), 'reads/s; cpu', c['value'], '; parity', c['parity_check']['reads'], 'reads', c['parity_check']['alignments'], 'alignments', c['parity_check']['mismatching_reads'], 'mismatching')"
done
