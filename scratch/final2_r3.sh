#!/bin/bash
# very end of round 3: records of the last tree (bash scratch/final2_r3.sh on the box)
set -e
mkdir -p gpurun_out/final2
timeout -k 10 300 python3 bench.py > gpurun_out/final2/r3_bench_default_i.json 2> gpurun_out/final2/default.err
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > gpurun_out/final2/r3_bench_driver_g.json 2> gpurun_out/final2/driver.err
for agg in max_pool mean_pool; do
  timeout -k 10 240 python3 bench.py --aggregator $agg --cpu-steps 0 --no-variant > gpurun_out/final2/r3_bench_${agg}_d.json 2> gpurun_out/final2/bench_$agg.err
done
timeout -k 10 240 python3 bench.py --aggregator lstm --cpu-steps 0 --no-variant --steps 100 --warmup 10 --settle-steps 20 > gpurun_out/final2/r3_bench_lstm_engine_e.json 2> gpurun_out/final2/lstm.err
timeout -k 10 500 python3 -m graphsage_amd.experiments.fig2a --dataset reddit-shaped --aggregators mean sum max gcn mean_pool max_pool lstm --epochs 3 > gpurun_out/final2/fig2a.txt 2> gpurun_out/final2/fig2a.err
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final2/smoke.log 2>&1
