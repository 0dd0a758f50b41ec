#!/bin/bash
# knock-outs of the max_pool edge pass (temporary SAGE_KO switch: 1 no atomics, 2 no own-row stores, 4 no arg-max loads)
for ko in 0 1 2 3 4 7; do
  SAGE_KO=$ko timeout -k 10 100 python -u scratch/pool_bwd_time.py 2>&1 | grep "max indeg" | sed "s/^/ko=$ko /" || exit 1
done
