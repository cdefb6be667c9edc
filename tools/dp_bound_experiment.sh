#!/bin/bash
# Which issue port bounds the extension kernel's diagonal loop?  Runs the 150 bp bench with the product library and with
# two experiment builds that add 10 scalar / 10 packed-VALU instructions per diagonal (make -C ma_amd/csrc exp).
WL=${1:-150bp} # 150bp: the extension kernel (ksw_ext.h); 10kb: the exact kernel (ksw_pk.h)
for v in "" salu valu; do
  lib=""; [ -n "$v" ] && lib=$PWD/tools/_prof/libma_amd_$v.so
  r=$(MA_AMD_LIB=$lib python bench.py --workload $WL --steps $([ $WL = 150bp ] && echo 12 || echo 2) --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['kernel_ms_per_step']['k_ksw'])")
  echo "variant=${v:-product} ms_per_step k_ksw_ms: $r"
done
