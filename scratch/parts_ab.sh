#!/bin/bash
# headline step: layer-0 weight gradients as stored split-K slices summed by Adam (default) against fp32 atomics (SAGE_NO_DW_PARTS=1)
for rep in 1 2 3; do
for v in "" 1; do
  SAGE_NO_DW_PARTS=$v timeout -k 10 200 python bench.py --steps 1000 --warmup 50 --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('SAGE_NO_DW_PARTS=%-2s' % '$v', 'overlapped', d['ms_per_step'], d['final_loss'])" || exit 1
done
done
