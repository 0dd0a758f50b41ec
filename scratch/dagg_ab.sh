# same-box A/B: layer-0 d agg as a forward-type bf16x3 contraction on W_l^T (default) against the exact-fp32 d-input kernel
set -e
for rep in 1 2; do
  for v in "" 1; do
    SAGE_NO_DAGG_X3=$v timeout -k 10 200 python bench.py --aggregator max_pool --cpu-steps 0 --no-variant --no-probe --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('max_pool  NO_DAGG_X3=%s' % '$v', d['ms_per_step'])"
    SAGE_NO_DAGG_X3=$v timeout -k 10 200 python bench.py --aggregator mean_pool --cpu-steps 0 --no-variant --no-probe --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('mean_pool NO_DAGG_X3=%s' % '$v', d['ms_per_step'])"
  done
done
SAGE_NO_DAGG_X3= timeout -k 10 200 python bench.py --aggregator lstm --cpu-steps 0 --no-variant --no-probe --steps 60 --warmup 10 --settle-steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('lstm default', d['ms_per_step'])"
SAGE_NO_DAGG_X3=1 timeout -k 10 200 python bench.py --aggregator lstm --cpu-steps 0 --no-variant --no-probe --steps 60 --warmup 10 --settle-steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('lstm NO_DAGG_X3=1', d['ms_per_step'])"
