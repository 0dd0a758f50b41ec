#!/bin/bash
# headline step: the next batch's gather gated by the previous step's end-of-step event (default)
# against a `started` event of its own at the top of every step (SAGE_STARTED_EVENT=1)
for rep in 1 2 3; do
for v in "" 1; do
  SAGE_STARTED_EVENT=$v timeout -k 10 200 python bench.py --steps 1000 --warmup 50 --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('SAGE_STARTED_EVENT=%-2s' % '$v', 'overlapped', d['ms_per_step'], d['final_loss'])" || exit 1
done
done
