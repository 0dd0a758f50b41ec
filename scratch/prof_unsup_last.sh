set -e
export TMPDIR=/tmp
R=/tmp/pul_raw; rm -rf $R; mkdir -p $R gpurun_out/pul
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/u -o t -- python3 bench.py --workload products --variant unsup --cpu-steps 0 --no-probe --steps 200 --settle-steps 50 > gpurun_out/pul/u.json 2> gpurun_out/pul/u.err
python3 profiles/timeline.py $R/u/t_kernel_trace.csv 3 > gpurun_out/pul/unsup.timeline.txt
cp $R/u/t_kernel_stats.csv gpurun_out/pul/unsup_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o t -- python3 bench.py --variant trainer --cpu-steps 0 --no-probe --steps 200 --settle-steps 50 > gpurun_out/pul/t.json 2> gpurun_out/pul/t.err
python3 profiles/timeline.py $R/t/t_kernel_trace.csv 3 > gpurun_out/pul/trainer.timeline.txt
cp $R/t/t_kernel_stats.csv gpurun_out/pul/trainer_kernel_stats.csv
