#!/bin/bash
# headline step: class-logit layer as fwd+bwd launch + rider on the hidden layer's pass (default)
# against forward / backward(+dW) launches apart (SAGE_NO_LOGIT_FWDBWD=1)
for rep in 1 2; do
for v in "" 1; do
  SAGE_NO_LOGIT_FWDBWD=$v timeout -k 10 200 python bench.py --steps 1000 --warmup 50 --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('SAGE_NO_LOGIT_FWDBWD=%-2s' % '$v', 'overlapped', d['ms_per_step'], d['final_loss'])" || exit 1
  SAGE_NO_LOGIT_FWDBWD=$v timeout -k 10 200 python bench.py --serial --steps 1000 --warmup 50 --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('SAGE_NO_LOGIT_FWDBWD=%-2s' % '$v', 'serial    ', d['ms_per_step'], d['final_loss'])" || exit 1
done
done
