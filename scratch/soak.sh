# longer runs of every engine aggregator on the last tree: the loss stays finite, nothing hangs
set -e
for spec in "mean 50000" "sum 20000" "max 20000" "gcn 20000" "max_pool 5000" "mean_pool 5000" "lstm 1000"; do
  set -- $spec
  timeout -k 10 300 python bench.py --aggregator $1 --cpu-steps 0 --no-variant --no-probe --steps $2 --warmup 10 --settle-steps 20 2>/dev/null | python3 -c "import sys,json,math; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); assert math.isfinite(d['final_loss']); print('$1', d['steps'], 'steps', d['ms_per_step'], 'ms/step, final loss', d['final_loss'])"
done
