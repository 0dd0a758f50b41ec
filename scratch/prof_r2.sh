#!/bin/bash
# Round-2 profile collection (run on the GPU box from the repo root):  bash scratch/prof_r2.sh
# kernel traces first, counters in passes of their own (never combined with other trace domains);
# raw rocprofv3 output stays in /tmp on the box, the summaries land in gpurun_out/r2prof
set -e
export TMPDIR=/tmp
R=/tmp/r2prof_raw
O=$PWD/gpurun_out/r2prof
rm -rf $R $O
mkdir -p $R $O
T="timeout -k 10 240"
$T rocprofv3 --kernel-trace --stats -d $R/head_ov -o head_ov -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --steps 500 > $O/head_ov.json 2> $O/head_ov.err
python3 profiles/db_kernel_stats.py $R/head_ov/head_ov_results.db $O/r2_bench_kernel_stats.csv > $O/r2_bench_kernel_stats.txt
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/head_serial -o head_serial -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 300 --settle-steps 300 > $O/head_serial.json 2> $O/head_serial.err
python3 profiles/timeline.py $R/head_serial/head_serial_kernel_trace.csv 3 > $O/r2_bench_step_timeline.txt
cp $R/head_serial/head_serial_kernel_stats.csv $O/r2_bench_serial_kernel_stats.csv
$T rocprofv3 --kernel-trace --stats -d $R/pool -o pool -- python3 bench.py --aggregator max_pool --steps 30 --warmup 5 --cpu-steps 0 > $O/pool.json 2> $O/pool.err
python3 profiles/db_kernel_stats.py $R/pool/pool_results.db $O/r2_pool_kernel_stats.csv > $O/r2_pool_kernel_stats.txt
$T rocprofv3 --kernel-trace --stats -d $R/lstm -o lstm -- python3 bench.py --aggregator lstm --steps 8 --warmup 2 --cpu-steps 0 > $O/lstm.json 2> $O/lstm.err
python3 profiles/db_kernel_stats.py $R/lstm/lstm_results.db $O/r2_lstm_kernel_stats.csv > $O/r2_lstm_kernel_stats.txt
echo traces done
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $R/pmc_pool -o pmc_pool -- python3 bench.py --aggregator max_pool --steps 4 --warmup 2 --cpu-steps 0 > $O/pmc_pool.json 2> $O/pmc_pool.err
python3 profiles/pmc_summary.py $R/pmc_pool/pmc_pool_counter_collection.csv gemm > $O/r2_pool_pmc_sq.txt
timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $R/pmc_lstm -o pmc_lstm -- python3 bench.py --aggregator lstm --steps 2 --warmup 1 --cpu-steps 0 > $O/pmc_lstm.json 2> $O/pmc_lstm.err
python3 profiles/pmc_summary.py $R/pmc_lstm/pmc_lstm_counter_collection.csv gemm > $O/r2_lstm_pmc_sq.txt
timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $R/pmc_head -o pmc_head -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 20 --warmup 5 --settle-steps 20 > $O/pmc_head.json 2> $O/pmc_head.err
python3 profiles/pmc_summary.py $R/pmc_head/pmc_head_counter_collection.csv gemm > $O/r2_gemm_pmc_sq.txt
python3 profiles/pmc_summary.py $R/pmc_head/pmc_head_counter_collection.csv gather_reduce_fwd > $O/r2_gather_pmc_sq.txt
echo pmc done
du -sh $R $O
