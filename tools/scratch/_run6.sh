export TMPDIR=/tmp
for wl in 50kb 10kb; do
rocprofv3 --kernel-trace -d gpurun_out/tr_$wl -o tr --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/tr_$wl.log 2>&1
python3 tools/launch_list.py gpurun_out/tr_$wl k_chain k_sort k_soc k_stitch k_hs k_seed > gpurun_out/r03b_launch_timeline_$wl.txt; rm -rf gpurun_out/tr_$wl
done
