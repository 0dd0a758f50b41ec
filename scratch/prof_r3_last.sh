#!/bin/bash
# last tree of round 3: kernel stats of the default bench command (overlapped) and the serial schedule
set -e
export TMPDIR=/tmp
R=/tmp/r3last_raw
O=$PWD/gpurun_out/r3last
rm -rf $R $O
mkdir -p $R $O
T="timeout -k 10 240"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/ov -o ov -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --steps 500 > $O/ov.json 2> $O/ov.err
cp $R/ov/ov_kernel_stats.csv $O/r3_bench_kernel_stats_b.csv
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/se -o se -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 300 --settle-steps 300 > $O/se.json 2> $O/se.err
cp $R/se/se_kernel_stats.csv $O/r3_bench_serial_kernel_stats_b.csv
