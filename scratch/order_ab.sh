# same-box A/B of the enqueue order inside pipelined_step (main stream first = default, SAGE_SIDE_FIRST=1 = round 2's order),
# in the driver's 20-step window and in the 1000-step default
for rep in 1 2 3; do
  for v in "" 1; do
    SAGE_SIDE_FIRST=$v timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('20 steps   SIDE_FIRST=%s' % '$v', d['ms_per_step'])"
  done
done
for v in "" 1; do
  SAGE_SIDE_FIRST=$v timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --no-probe 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('1000 steps SIDE_FIRST=%s' % '$v', d['ms_per_step'])"
done
