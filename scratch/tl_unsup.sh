set -e
export TMPDIR=/tmp
R=/tmp/tlu_raw; rm -rf $R; mkdir -p $R gpurun_out/tlu
timeout -k 10 300 python3 bench.py --workload products --variant unsup --cpu-steps 0 --no-probe --steps 300 > gpurun_out/tlu/plain.json 2> gpurun_out/tlu/plain.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o t -- python3 bench.py --workload products --variant unsup --cpu-steps 0 --no-probe --steps 100 --settle-steps 50 > gpurun_out/tlu/t.json 2> gpurun_out/tlu/t.err
python3 profiles/timeline.py $R/t/t_kernel_trace.csv 3 > gpurun_out/tlu/timeline.txt
