set -e
export TMPDIR=/tmp
R=/tmp/tlp_raw; rm -rf $R; mkdir -p $R gpurun_out/tlp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/p -o t -- python3 bench.py --workload ppi --cpu-steps 0 --no-probe --no-variant --steps 200 --settle-steps 50 > gpurun_out/tlp/p.json 2> gpurun_out/tlp/p.err
python3 profiles/timeline.py $R/p/t_kernel_trace.csv 3 > gpurun_out/tlp/ppi.timeline.txt
cp $R/p/t_kernel_stats.csv gpurun_out/tlp/ppi_kernel_stats.csv
