# same-box A/B: layer 0's pooling edge pass pulled by the source rows (default) against pushed from the targets (SAGE_NO_POOL_PULL=1)
set -e
for rep in 1 2; do
  for v in "" 1; do
    for agg in max_pool mean_pool; do
      SAGE_NO_POOL_PULL=$v timeout -k 10 200 python bench.py --aggregator $agg --cpu-steps 0 --no-variant --no-probe --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$agg NO_POOL_PULL=%s' % '$v', d['ms_per_step'])"
    done
  done
done
