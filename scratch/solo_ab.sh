#!/bin/bash
# headline step: hidden-gradient rows with one writer stored (default) against every row added atomically (SAGE_NO_SOLO_ROWS=1)
for rep in 1 2 3; do
for v in "" 1; do
  SAGE_NO_SOLO_ROWS=$v timeout -k 10 200 python bench.py --steps 1000 --warmup 50 --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('SAGE_NO_SOLO_ROWS=%-2s' % '$v', 'overlapped', d['ms_per_step'], d['final_loss'])" || exit 1
done
done
