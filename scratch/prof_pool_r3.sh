#!/bin/bash
# pooling steps with the pulled edge pass: kernel traces + timelines (bash scratch/prof_pool_r3.sh on the box)
set -e
export TMPDIR=/tmp
R=/tmp/r3pp_raw
O=$PWD/gpurun_out/r3pp
rm -rf $R $O
mkdir -p $R $O
T="timeout -k 10 240"
for agg in max_pool mean_pool; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$agg -o pe -- python3 bench.py --aggregator $agg --steps 60 --warmup 10 --settle-steps 30 --cpu-steps 0 --no-probe --no-variant > $O/$agg.json 2> $O/$agg.err
  cp $R/$agg/pe_kernel_stats.csv $O/r3_${agg}_engine_kernel_stats.csv
  python3 profiles/timeline.py $R/$agg/pe_kernel_trace.csv 3 > $O/r3_${agg}_engine_step_timeline.txt
done
