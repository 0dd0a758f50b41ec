#!/bin/bash
# kernel trace of one informational variant:  bash scratch/prof_variant.sh <name> <bench args...>
set -e
export TMPDIR=/tmp
N=$1; shift
R=/tmp/var_raw_$N; O=$PWD/gpurun_out/var_$N
rm -rf $R $O; mkdir -p $R $O
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/t -o t -- python3 bench.py "$@" --steps 100 --warmup 10 --settle-steps 50 --cpu-steps 0 --no-probe --no-variant --serial > $O/t.json 2> $O/t.err
python3 profiles/timeline.py $R/t/t_kernel_trace.csv 3 > $O/timeline.txt
