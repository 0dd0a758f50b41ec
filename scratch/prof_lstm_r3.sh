#!/bin/bash
# lstm engine step: kernel trace + SQ counter pass (run on the GPU box from the repo root)
set -e
export TMPDIR=/tmp
R=/tmp/r3lstm_raw
O=$PWD/gpurun_out/r3lstm
rm -rf $R $O
mkdir -p $R $O
T="timeout -k 10 240"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/lstm -o lstm -- python3 bench.py --aggregator lstm --steps 8 --warmup 2 --settle-steps 4 --cpu-steps 0 --no-variant --no-probe > $O/lstm.json 2> $O/lstm.err
python3 profiles/timeline.py $R/lstm/lstm_kernel_trace.csv 3 > $O/r3_lstm_engine_step_timeline.txt
cp $R/lstm/lstm_kernel_stats.csv $O/r3_lstm_engine_kernel_stats.csv
echo traces done
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $R/pmc_lstm -o pmc_lstm -- python3 bench.py --aggregator lstm --steps 2 --warmup 1 --settle-steps 1 --cpu-steps 0 --no-variant --no-probe > $O/pmc_lstm.json 2> $O/pmc_lstm.err
python3 profiles/pmc_summary.py $R/pmc_lstm/pmc_lstm_counter_collection.csv gemm > $O/r3_lstm_engine_pmc_sq.txt
echo pmc done
