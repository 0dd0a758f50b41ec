#!/bin/bash
# lstm step (module path) at the Reddit shape: bench line + kernel stats (run on the GPU box)
set -e
export TMPDIR=/tmp
R=/tmp/lstm_raw; O=$PWD/gpurun_out/lstm_prof
rm -rf $R $O; mkdir -p $R $O
timeout -k 10 280 python3 bench.py --aggregator lstm --steps 20 --warmup 3 --cpu-steps 0 --no-variant > $O/r2_bench_lstm.json 2> $O/bench.err
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $R/lstm -o lstm -- python3 bench.py --aggregator lstm --steps 8 --warmup 2 --cpu-steps 0 > $O/lstm.json 2> $O/lstm.err
python3 profiles/db_kernel_stats.py $R/lstm/lstm_results.db $O/r2_lstm_kernel_stats.csv > $O/r2_lstm_kernel_stats.txt
