#!/bin/bash
# Where the host-to-host leg of the 150 bp workload loses against the device-resident one: the same leg with one direction only,
# double-buffered (the default) or serial, with more batches in flight, with the copies forced onto blit kernels / with event waits.  One line per variant into
# gpurun_out/h2h_experiment.txt.   usage (GPU box): bash tools/h2h_experiment.sh [steps]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
K=${1:-30}
OUT=gpurun_out/h2h_experiment.txt
: > $OUT
run() {
    tag=$1; nfl=$2; hio=$3; shift 3
    env "$@" python bench.py --workload 150bp --steps $K --warmup 2 --cpu-sample 0 --boundary-reads 0 --inflight $nfl --host-io $hio \
        --detail-file gpurun_out/h2h_$tag.json > gpurun_out/h2h_$tag.line 2> gpurun_out/h2h_$tag.err
    python - "$tag" >> $OUT <<'P'
import json, sys
tag = sys.argv[1]
try:
    d = json.load(open("gpurun_out/h2h_%s.json" % tag))["workloads"][0]
    print("%-22s %12.1f reads/s  %7.3f ms/step  phases %s" % (tag, d.get("value", 0), d.get("ms_per_step", 0), d.get("host_phases_ms_per_step")))
except Exception as e:
    print("%-22s failed: %r" % (tag, e))
P
}
run device_resident_3 3 0 MA_X=0
run h2h_3 3 1 MA_X=0
run h2h_3_serial 3 1 MA_BENCH_H2H_SERIAL=1
run h2h_2 2 1 MA_X=0
run h2h_4 4 1 MA_X=0
run h2h_3_upload_only 3 1 MA_BENCH_H2H=up
run h2h_3_download_only 3 1 MA_BENCH_H2H=down
run h2h_3_no_sdma 3 1 HSA_ENABLE_SDMA=0
run h2h_3_event_waits 3 1 MA_BENCH_BLOCKING_SYNC=1
# where the host threads run (ma_host_bind_thread; the default of bench.py is the GPU's own CPUs)
run h2h_3_remote_cpus 3 1 MA_BENCH_BIND=remote
run h2h_3_unpinned 3 1 MA_BENCH_BIND=none
run single_stream 1 0 MA_X=0
cat $OUT
