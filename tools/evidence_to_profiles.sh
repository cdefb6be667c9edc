#!/bin/bash
# Copies what tools/round_evidence.sh left under gpurun_out/ into profiles/ (tracked).  usage: bash tools/evidence_to_profiles.sh [tag=r03]
TAG=${1:-r06}
cd "$(dirname "$0")/.."
for f in parity_sweep.txt bench_default.json bench_detail.json bench_illumina.json gpu_tests.txt launch_timeline_50kb.txt launch_timeline_10kb.txt launch_timeline_50kb_nanopore.txt sq_counters_10kb_dp.txt ext_pairing_bound.txt overlap_timeline_150bp_h2h.txt kernel_stats_150bp_h2h_inflight3.csv bench_150bp_h2h_inflight3_under_rocprof.json overlap_matrix_150bp.txt overlap_matrix_150bp_cu_split.txt pk_phase_profile_10kb.txt step_timeline_150bp.txt grp_phase_profile_150bp.txt calibration.json binding_graph_rate.txt h2h_experiment.txt dp_job_histogram.txt band_soak.txt chain_wave_ab.txt chain_phase_profile.txt; do
  [ -s gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
[ -s gpurun_out/prof_${TAG}_pmc_traffic.json ] && cp gpurun_out/prof_${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
for d in 150bp 10kb 50kb 150bp_illumina 10kb_pacbio 50kb_nanopore; do
  [ -s gpurun_out/prof_${TAG}_$d/kernel_stats.csv ] && cp gpurun_out/prof_${TAG}_$d/kernel_stats.csv profiles/${TAG}_kernel_stats_$d.csv
  [ -s gpurun_out/prof_${TAG}_$d/summary.txt ] && cp gpurun_out/prof_${TAG}_$d/summary.txt profiles/${TAG}_pmc_summary_$d.txt
done
ls -la profiles | grep ${TAG}_
