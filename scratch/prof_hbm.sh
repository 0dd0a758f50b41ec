#!/bin/bash
# HBM bytes per launch from the TCC counters, one counter per pass (MI355X_MICROARCH.md, HBM):
#   bash scratch/prof_hbm.sh       (on the GPU box, from the repo root; summaries in gpurun_out/r2hbm)
set -e
export TMPDIR=/tmp
R=/tmp/r2hbm_raw
O=$PWD/gpurun_out/r2hbm
rm -rf $R $O
mkdir -p $R $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/$c -o p -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 20 --warmup 5 --settle-steps 20 > $O/$c.json 2> $O/$c.err
  python3 profiles/pmc_summary.py $R/$c/p_counter_collection.csv > $O/$c.txt
done
cat $O/FETCH_SIZE.txt $O/WRITE_SIZE.txt > $O/r2_bench_pmc_hbm.txt
