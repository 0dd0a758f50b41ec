#!/bin/bash
# knock-outs of the pushed max_pool edge pass (profiles/r3_pool_pull_ab.txt).  SAGE_KO was a TEMPORARY switch in a scratch
# build of gather.hip (an `int ko` kernel argument: 1 skip the atomics of shared rows, 2 skip the plain stores of
# singly-owned rows, 4 take the arg-max from arithmetic instead of loading it); the tree does not have it.
for ko in 0 1 2 3 4 7; do
  SAGE_KO=$ko timeout -k 10 100 python -u scratch/pool_bwd_time.py 2>&1 | grep "max indeg" | sed "s/^/ko=$ko /" || exit 1
done
