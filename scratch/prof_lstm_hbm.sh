#!/bin/bash
# lstm engine step: HBM fetch bytes and L2 hit / miss per kernel, one counter group per pass
set -e
export TMPDIR=/tmp
R=/tmp/r3lstmhbm_raw
O=$PWD/gpurun_out/r3lstmhbm
rm -rf $R $O
mkdir -p $R $O
i=0
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/p$i -o p -- python3 bench.py --aggregator lstm --steps 2 --warmup 1 --settle-steps 1 --cpu-steps 0 --no-variant --no-probe > $O/p$i.json 2> $O/p$i.err || { tail -5 $O/p$i.err; continue; }
  python3 profiles/pmc_summary.py $R/p$i/p_counter_collection.csv gemm > $O/p$i.txt
done
cat $O/p*.txt > $O/r3_lstm_engine_pmc_tcc.txt
