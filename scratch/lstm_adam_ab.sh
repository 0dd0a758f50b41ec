# same-box A/B: lstm step, Adam over the ranges that can have a gradient (default) against the whole flat buffer (SAGE_LSTM_ADAM_ALL=1)
set -e
for rep in 1 2; do
  for v in "" 1; do
    SAGE_LSTM_ADAM_ALL=$v timeout -k 10 200 python bench.py --aggregator lstm --cpu-steps 0 --no-variant --no-probe --steps 100 --warmup 10 --settle-steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('lstm ADAM_ALL=%s' % '$v', d['ms_per_step'])"
  done
done
