#!/bin/bash
# Round-3 profile collection (run on the GPU box from the repo root):  bash scratch/prof_r3.sh
# kernel traces only (the kernels' counter passes are round 2's: the contraction / gather kernels are unchanged);
# raw rocprofv3 output stays in /tmp on the box, the summaries land in gpurun_out/r3prof
set -e
export TMPDIR=/tmp
R=/tmp/r3prof_raw
O=$PWD/gpurun_out/r3prof
rm -rf $R $O
mkdir -p $R $O
T="timeout -k 10 240"
$T rocprofv3 --kernel-trace --stats -d $R/head_ov -o head_ov -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --steps 500 > $O/head_ov.json 2> $O/head_ov.err
python3 profiles/db_kernel_stats.py $R/head_ov/head_ov_results.db $O/r3_bench_kernel_stats.csv > $O/r3_bench_kernel_stats.txt
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/head_serial -o head_serial -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 300 --settle-steps 300 > $O/head_serial.json 2> $O/head_serial.err
python3 profiles/timeline.py $R/head_serial/head_serial_kernel_trace.csv 3 > $O/r3_bench_step_timeline.txt
cp $R/head_serial/head_serial_kernel_stats.csv $O/r3_bench_serial_kernel_stats.csv
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/lstm -o lstm -- python3 bench.py --aggregator lstm --steps 8 --warmup 2 --settle-steps 4 --cpu-steps 0 --no-variant --no-probe > $O/lstm.json 2> $O/lstm.err
python3 profiles/timeline.py $R/lstm/lstm_kernel_trace.csv 3 > $O/r3_lstm_engine_step_timeline.txt
cp $R/lstm/lstm_kernel_stats.csv $O/r3_lstm_engine_kernel_stats.csv
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/eval -o eval -- python3 scratch/eval_pass.py mean 1 > $O/eval_pass.txt 2> $O/eval.err
cp $R/eval/eval_kernel_stats.csv $O/r3_eval_pass_kernel_stats.csv
echo traces done
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $R/pmc_lstm -o pmc_lstm -- python3 bench.py --aggregator lstm --steps 2 --warmup 1 --settle-steps 1 --cpu-steps 0 --no-variant --no-probe > $O/pmc_lstm.json 2> $O/pmc_lstm.err
python3 profiles/pmc_summary.py $R/pmc_lstm/pmc_lstm_counter_collection.csv gemm > $O/r3_lstm_engine_pmc_sq.txt
echo pmc done
du -sh $R $O
