#!/bin/bash
# Round-2 profile refresh after the fused logit layer, the pooling backward rework and the
# XCD-aware tile order (run on the GPU box from the repo root):  bash scratch/prof_r2b.sh
# raw rocprofv3 output stays in /tmp on the box, the summaries land in gpurun_out/r2prof_b
set -e
export TMPDIR=/tmp
R=/tmp/r2prof_raw
O=$PWD/gpurun_out/r2prof_b
rm -rf $R $O
mkdir -p $R $O
T="timeout -k 10 240"
$T python3 bench.py > $O/r2_bench_default_b.json 2> $O/default.err
echo default done
$T rocprofv3 --kernel-trace --stats -d $R/head_ov -o head_ov -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --steps 500 > $O/head_ov.json 2> $O/head_ov.err
python3 profiles/db_kernel_stats.py $R/head_ov/head_ov_results.db $O/r2_bench_kernel_stats.csv > $O/r2_bench_kernel_stats.txt
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/head_serial -o head_serial -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 300 --settle-steps 300 > $O/head_serial.json 2> $O/head_serial.err
python3 profiles/timeline.py $R/head_serial/head_serial_kernel_trace.csv 3 > $O/r2_bench_step_timeline.txt
cp $R/head_serial/head_serial_kernel_stats.csv $O/r2_bench_serial_kernel_stats.csv
for agg in max_pool mean_pool; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$agg -o pe -- python3 bench.py --aggregator $agg --steps 60 --warmup 10 --settle-steps 30 --cpu-steps 0 --no-probe --no-variant > $O/$agg.json 2> $O/$agg.err
  cp $R/$agg/pe_kernel_stats.csv $O/r2_${agg}_engine_kernel_stats.csv
  python3 profiles/timeline.py $R/$agg/pe_kernel_trace.csv 3 > $O/r2_${agg}_engine_step_timeline.txt
  $T python3 bench.py --aggregator $agg --cpu-steps 0 --no-variant > $O/r2_bench_${agg}.json 2> $O/bench_$agg.err
done
echo traces done
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $R/pmc_head -o pmc_head -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 20 --warmup 5 --settle-steps 20 > $O/pmc_head.json 2> $O/pmc_head.err
python3 profiles/pmc_summary.py $R/pmc_head/pmc_head_counter_collection.csv gemm > $O/r2_gemm_pmc_sq.txt
python3 profiles/pmc_summary.py $R/pmc_head/pmc_head_counter_collection.csv logit > $O/r2_logit_pmc_sq.txt
timeout -s KILL 200 rocprofv3 --pmc $PMC --output-format csv -d $R/pmc_pool -o pmc_pool -- python3 bench.py --aggregator max_pool --steps 4 --warmup 2 --settle-steps 4 --cpu-steps 0 --no-probe --no-variant > $O/pmc_pool.json 2> $O/pmc_pool.err
python3 profiles/pmc_summary.py $R/pmc_pool/pmc_pool_counter_collection.csv gemm > $O/r2_pool_engine_pmc_sq.txt
echo pmc done
du -sh $R $O
