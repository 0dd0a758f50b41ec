# kernel timeline of one step in the data-parallel control flow (1-rank RCCL group) against the serial schedule
set -e
export TMPDIR=/tmp
R=/tmp/tl_raw; rm -rf $R; mkdir -p $R gpurun_out/tlc
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/c -o c -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --force-collective --steps 200 --settle-steps 100 > gpurun_out/tlc/c.json 2> gpurun_out/tlc/c.err
python3 profiles/timeline.py $R/c/c_kernel_trace.csv 3 > gpurun_out/tlc/timeline_coll.txt
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/s -o s -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 200 --settle-steps 100 > gpurun_out/tlc/s.json 2> gpurun_out/tlc/s.err
python3 profiles/timeline.py $R/s/s_kernel_trace.csv 3 > gpurun_out/tlc/timeline_serial.txt
