#!/bin/bash
# pooling edge pass, pushed and pulled: HBM fetch / write bytes per launch on one Reddit-shaped batch (scratch/pool_bwd_time.py), one counter per pass
set -e
export TMPDIR=/tmp
R=/tmp/r3poolhbm_raw
O=$PWD/gpurun_out/r3poolhbm
rm -rf $R $O
mkdir -p $R $O
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/p$i -o p -- python3 scratch/pool_bwd_time.py > $O/p$i.log 2> $O/p$i.err || { tail -5 $O/p$i.err; continue; }
  python3 profiles/pmc_summary.py $R/p$i/p_counter_collection.csv pool_ > $O/p$i.txt
done
cat $O/p*.txt > $O/r3_pool_edge_pass_pmc_tcc.txt
