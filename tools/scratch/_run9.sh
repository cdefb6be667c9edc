export TMPDIR=/tmp
python bench.py --workload 50kb --steps 3 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/ab_50kb_conc.json 2> gpurun_out/ab_50kb_conc.err
MA_DP_ONE_STREAM=1 python bench.py --workload 50kb --steps 3 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/ab_50kb_one.json 2> gpurun_out/ab_50kb_one.err
rocprofv3 --kernel-trace -d gpurun_out/tr_50kb -o tr --output-format csv -- python3 bench.py --workload 50kb --steps 1 --warmup 1 --cpu-sample 0 --boundary-reads 0 --overlap 0 > gpurun_out/tr_50kb.log 2>&1
python3 tools/launch_list.py gpurun_out/tr_50kb k_ksw k_job k_stitch > gpurun_out/r03c_launch_timeline_50kb.txt; rm -rf gpurun_out/tr_50kb
rocm-smi --showcomputepartition --showmemorypartition > gpurun_out/smi.txt 2>&1
