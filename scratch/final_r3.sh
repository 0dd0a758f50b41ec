#!/bin/bash
# end of round 3: the records of the final tree (bash scratch/final_r3.sh on the box)
set -e
mkdir -p gpurun_out/final
timeout -k 10 300 python3 bench.py > gpurun_out/final/r3_bench_default_h.json 2> gpurun_out/final/default.err
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > gpurun_out/final/r3_bench_driver_f.json 2> gpurun_out/final/driver.err
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1
