#!/bin/bash
# headline step: stream-K forward (default) against whole tiles without the atomic epilogue
# (SAGE_FWD_WHOLE=64: 410 tiles of 64 x 128 on 768 slots; =128: 206 tiles of 128 x 128 on 512)
for rep in 1 2; do
for v in "" 64 128; do
  SAGE_FWD_WHOLE=$v timeout -k 10 200 python bench.py --steps 1000 --warmup 50 --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('SAGE_FWD_WHOLE=%-4s' % '$v', 'overlapped', d['ms_per_step'], d['final_loss'])" || exit 1
  SAGE_FWD_WHOLE=$v timeout -k 10 200 python bench.py --serial --steps 1000 --warmup 50 --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('SAGE_FWD_WHOLE=%-4s' % '$v', 'serial    ', d['ms_per_step'], d['final_loss'])" || exit 1
done
done
