#!/bin/bash
# pool steps after the forked weight gradients: bench lines + kernel traces (bash scratch/prof_pool_fork.sh on the box)
set -e
export TMPDIR=/tmp
R=/tmp/r2pf_raw
O=$PWD/gpurun_out/r2pf
rm -rf $R $O
mkdir -p $R $O
T="timeout -k 10 240"
for agg in max_pool mean_pool; do
  $T python3 bench.py --aggregator $agg --cpu-steps 0 --no-variant > $O/r2_bench_${agg}.json 2> $O/bench_$agg.err
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$agg -o pe -- python3 bench.py --aggregator $agg --steps 60 --warmup 10 --settle-steps 30 --cpu-steps 0 --no-probe --no-variant > $O/$agg.json 2> $O/$agg.err
  cp $R/$agg/pe_kernel_stats.csv $O/r2_${agg}_engine_kernel_stats.csv
  python3 profiles/timeline.py $R/$agg/pe_kernel_trace.csv 3 > $O/r2_${agg}_engine_step_timeline.txt
done
