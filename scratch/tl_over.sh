set -e
export TMPDIR=/tmp
R=/tmp/tlo_raw; rm -rf $R; mkdir -p $R gpurun_out/tlo
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o t -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --steps 400 --settle-steps 100 > gpurun_out/tlo/t.json 2> gpurun_out/tlo/t.err
python3 profiles/timeline.py $R/t/t_kernel_trace.csv 3 > gpurun_out/tlo/timeline.txt
python3 profiles/timeline.py $R/t/t_kernel_trace.csv 150 > gpurun_out/tlo/timeline_mid.txt
python3 scratch/tl_stats.py $R/t/t_kernel_trace.csv > gpurun_out/tlo/stats.txt
