#!/bin/bash
# Everything the round's committed numbers come from, in one GPU call (≈38 min of box time in round 6; the part before QUICK's exit ≈23 min): parity tests, rocprofv3
# kernel stats + PMC passes of the four workloads, the bench line (all workloads, three legs each, C1 anchor, boundary), the
# kernel trace of the leg `value` comes from and its concurrency timeline, the overlap matrix, the phase profile of
# k_ksw_pk<5>, SQ counters of the 10 kb DP stage, launch timelines.  tools/evidence_to_profiles.sh copies what is to be judged
# from gpurun_out/ into profiles/ (profiles/README.md).
#   usage: bash tools/round_evidence.sh [tag=r06]
TAG=${1:-r06}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/${TAG}_gpu_tests.txt
cat gpurun_out/${TAG}_gpu_tests.txt
# the PMC passes first: bench.py replays their HBM bytes / VALU instructions (roofline.traffic, wave_insts_per_launch) only
# when they were collected with the kernel sources the library is built from (kernel_source_hash)
for wl in 150bp 10kb 50kb; do bash tools/collect_profiles.sh $TAG $wl > gpurun_out/collect_$wl.log 2>&1; done
bash tools/collect_profiles.sh $TAG 150bp illumina > gpurun_out/collect_illumina.log 2>&1
# (round 6) the legs under the reference's PacBio and Nanopore parameter sets
for wl in 10kb_pacbio 50kb_nanopore; do bash tools/collect_profiles.sh $TAG $wl > gpurun_out/collect_$wl.log 2>&1; done
for wl in 150bp 10kb 50kb 150bp_illumina 10kb_pacbio 50kb_nanopore; do cp gpurun_out/prof_${TAG}_$wl/summary.txt gpurun_out/${TAG}_pmc_summary_$wl.txt 2>/dev/null; cp gpurun_out/prof_${TAG}_$wl/kernel_stats.csv gpurun_out/${TAG}_kernel_stats_$wl.csv 2>/dev/null; done
cp gpurun_out/prof_${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
# the ceilings the roofline fractions are quoted against, re-measured beside this build (tools/calibrate.sh without its PMC passes)
mkdir -p gpurun_out/calib_$TAG tools/_prof
hipcc --offload-arch=gfx950 -O3 tools/valu_mix.hip -o tools/_prof/valu_mix 2>/dev/null; hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result tools/gups.hip -o tools/_prof/gups 2>/dev/null
tools/_prof/valu_mix > gpurun_out/calib_$TAG/valu_mix.txt 2>&1; tools/_prof/gups > gpurun_out/calib_$TAG/gups.txt 2>&1
python3 tools/calibration_json.py gpurun_out/calib_$TAG > gpurun_out/${TAG}_calibration.json
cp gpurun_out/${TAG}_calibration.json profiles/${TAG}_calibration.json
# the driver's line (all workloads incl. the Illumina preset and the C1 anchor); its per-workload blocks go to the detail file
# (the driver's own command line)
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
echo "python bench.py --gpus 1 --steps 20 --warmup 5: $(( $(date +%s) - T0 )) s wall" > gpurun_out/${TAG}_bench_default_wall.txt
[ -n "$QUICK" ] && exit 0   # QUICK=1: only what depends on the kernel sources' hash (tests, PMC passes, calibration, the bench line)
make -s -C ma_amd/csrc prof > /dev/null 2>&1   # the diagnostics build of THESE sources (phase profiles below)
# where the host-to-host leg stands against the device-resident one: one direction only, serial instead of double-buffered I/O,
# copies forced onto blit kernels, host threads on the other socket
bash tools/h2h_experiment.sh 30 > /dev/null 2>&1; cp gpurun_out/h2h_experiment.txt gpurun_out/${TAG}_h2h_experiment.txt
# the proven narrow band against the oracle's kswcpp at the full band: 100 000 jobs, proved or handed on
python3 tools/band_soak.py 10 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/${TAG}_band_soak.txt
# the shapes of the kswcpp calls of the long-read workloads
( python3 tools/dp_job_histogram.py 150 200000 0.005 0 0; python3 tools/dp_job_histogram.py 10000 4000 0.004 0.003 0.003; python3 tools/dp_job_histogram.py 50000 1000 0.03 0.03 0.04 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_dp_job_histogram.txt
# the leg `value` comes from -- 150 bp, 3 batches in flight, host to host -- under rocprofv3 (program directly after --):
# kernel trace -> concurrency timeline, and the same command's --stats summary
rocprofv3 --kernel-trace --stats -d gpurun_out/tr_h2h -o tr --output-format csv -- python3 bench.py --workload 150bp --inflight 3 --host-io 1 --steps 9 --warmup 1 --cpu-sample 0 --boundary-reads 0 > gpurun_out/tr_h2h.log 2>&1
python3 tools/overlap_timeline.py $(find gpurun_out/tr_h2h -name "*kernel_trace.csv" | head -1) 3 1 > gpurun_out/${TAG}_overlap_timeline_150bp_h2h.txt
find gpurun_out/tr_h2h -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_150bp_h2h_inflight3.csv \;
grep "^{" gpurun_out/tr_h2h.log | tail -1 > gpurun_out/${TAG}_bench_150bp_h2h_inflight3_under_rocprof.json; rm -rf gpurun_out/tr_h2h
# the single-stream step launch by launch, and the phase profile of the kernel that runs several short extensions per wavefront
rocprofv3 --kernel-trace -d gpurun_out/tr150 -o tr --output-format csv -- python3 bench.py --workload 150bp --steps 3 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/tr150.log 2>&1
python3 tools/step_timeline.py gpurun_out/tr150 > gpurun_out/${TAG}_step_timeline_150bp.txt; rm -rf gpurun_out/tr150
python3 tools/grp_prof.py --workload 150bp --overlap 0 --boundary-reads 0 2>/dev/null | grep "^G =" > gpurun_out/${TAG}_grp_phase_profile_150bp.txt
python3 tools/binding_graph_rate.py > gpurun_out/${TAG}_binding_graph_rate.txt 2>&1
python3 tools/pk_prof.py --workload 10kb 2>&1 | grep -v "^{" | grep -v "^bench detail:" | grep -v amdgpu.ids > gpurun_out/${TAG}_pk_phase_profile_10kb.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU \
  -d gpurun_out/sq10 -o pmc --output-format csv -- python3 bench.py --workload 10kb --steps 1 --warmup 0 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/sq10.log 2>&1
python3 tools/pmc_sq.py gpurun_out/sq10 k_ksw > gpurun_out/${TAG}_sq_counters_10kb_dp.txt; rm -rf gpurun_out/sq10
for wl in 50kb 10kb 50kb_nanopore; do
rocprofv3 --kernel-trace -d gpurun_out/tr_$wl -o tr --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/tr_$wl.log 2>&1
python3 tools/launch_list.py gpurun_out/tr_$wl k_ksw k_job_cost k_chain k_sort_seeds k_soc_windows k_stitch k_seed k_task k_dp_enum > gpurun_out/${TAG}_launch_timeline_$wl.txt; rm -rf gpurun_out/tr_$wl
done
# the chain stage of the long-read workloads: its wave kernels on / off, and the phases of what is left one read per lane
bash tools/chain_ab.sh > gpurun_out/${TAG}_chain_wave_ab.txt 2>&1
make -s -C ma_amd/csrc chainprof > /dev/null 2>&1
for wl in 50kb_nanopore 50kb 10kb; do echo "== $wl"; python3 tools/chain_prof.py --workload $wl 2>&1 | grep -v "^{" | grep -v amdgpu.ids | grep -v "^bench detail" | tail -11; done > gpurun_out/${TAG}_chain_phase_profile.txt
python3 tools/ksw_prof.py --workload 150bp --boundary-reads 0 --overlap 0 2>&1 | grep -v "^{" | grep -v "^bench detail:" | grep -v amdgpu.ids > gpurun_out/${TAG}_ext_pairing_bound.txt
python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], d["config"]["value_is"])
for k in sorted(d):
    if k.startswith(("value_", "parity_", "c1_", "cpu_reference_", "dropin_", "roofline_")):
        print(k, d[k])
PY
