#!/bin/bash
# end of round 3, pooling aggregators: bench lines + the sweep's timing (bash scratch/pool_r3_final.sh on the box)
set -e
mkdir -p gpurun_out/r3pf
for agg in max_pool mean_pool; do
  timeout -k 10 240 python3 bench.py --aggregator $agg --cpu-steps 0 --no-variant > gpurun_out/r3pf/r3_bench_${agg}_c.json 2> gpurun_out/r3pf/bench_$agg.err
done
timeout -k 10 400 python3 -m graphsage_amd.experiments.fig2a --dataset reddit-shaped --aggregators mean max_pool mean_pool --epochs 2 > gpurun_out/r3pf/fig2a_pool.txt 2> gpurun_out/r3pf/fig2a_pool.err
