#!/bin/bash
# kernel trace of the mean_pool step on the engine (raw output stays in /tmp on the box)
set -e
export TMPDIR=/tmp
R=/tmp/r2mpool_raw; O=$PWD/gpurun_out/r2mpool
rm -rf $R $O; mkdir -p $R $O
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/pe -o pe -- python3 bench.py --aggregator mean_pool --steps 60 --warmup 10 --settle-steps 30 --cpu-steps 0 --no-probe --no-variant > $O/pe.json 2> $O/pe.err
cp $R/pe/pe_kernel_stats.csv $O/r2_pool_engine_kernel_stats.csv
python3 profiles/timeline.py $R/pe/pe_kernel_trace.csv 3 > $O/r2_pool_engine_step_timeline.txt
