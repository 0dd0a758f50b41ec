# end of round 3, one box: the three schedules of the headline step (overlapped, serial, data-parallel control flow on a 1-rank RCCL group)
set -e
for rep in 1 2; do
  timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --no-probe --steps 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('overlapped', d['ms_per_step'])"
  timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('serial', d['ms_per_step'])"
  timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --no-probe --force-collective --steps 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('data-parallel control flow (1 rank)', d['ms_per_step'], d.get('schedule'))"
done
