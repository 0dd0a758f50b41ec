# same-box A/B: max and max_pool steps with their class-logit layer on four waves per seed (default) against one (SAGE_LOGIT_ONE_WAVE=1)
set -e
for rep in 1 2; do
  for v in "" 1; do
    if [ -n "$v" ]; then export SAGE_LOGIT_ONE_WAVE=1; else unset SAGE_LOGIT_ONE_WAVE; fi
    timeout -k 10 200 python bench.py --aggregator max --cpu-steps 0 --no-variant --no-probe --steps 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('max ONE_WAVE=%s' % '$v', d['ms_per_step'])"
    timeout -k 10 200 python bench.py --aggregator max_pool --cpu-steps 0 --no-variant --no-probe --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('max_pool ONE_WAVE=%s' % '$v', d['ms_per_step'])"
  done
done
