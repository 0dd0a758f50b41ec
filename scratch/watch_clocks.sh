#!/bin/bash
# Run a GPU load in the background and sample the shader clock beside it (rocm-smi, every 0.5 s).
# usage: scratch/watch_clocks.sh <samples> <command ...>     e.g.  REPS=45000 scratch/watch_clocks.sh 4 scratch/bf16x3_gemm
n=$1; shift
"$@" > gpurun_out/watch_clocks_cmd.txt 2>&1 &
pid=$!
sleep 0.7
for i in $(seq $n); do
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
  sleep 0.5
done
wait $pid
tail -1 gpurun_out/watch_clocks_cmd.txt | cut -c1-300
