# same-box A/B: mean_pool, pulled edge pass for the layers above layer 0 (default) against the all-atomic pushed pass there (SAGE_NO_POOL_PULL_UP=1)
set -e
for rep in 1 2; do
  for v in "" 1; do
    SAGE_NO_POOL_PULL_UP=$v timeout -k 10 200 python bench.py --aggregator mean_pool --cpu-steps 0 --no-variant --no-probe --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('mean_pool NO_POOL_PULL_UP=%s' % '$v', d['ms_per_step'])"
  done
done
