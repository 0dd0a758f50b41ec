#!/bin/bash
# one serial step of the headline as a kernel timeline (run on the GPU box from the repo root)
set -e
export TMPDIR=/tmp
R=/tmp/tlhead_raw
rm -rf $R; mkdir -p $R gpurun_out
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/h -o h -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 300 --settle-steps 300 > gpurun_out/tl_head.json 2> gpurun_out/tl_head.err
python3 profiles/timeline.py $R/h/h_kernel_trace.csv 3 > gpurun_out/tl_head.txt
cp $R/h/h_kernel_stats.csv gpurun_out/tl_head_stats.csv
