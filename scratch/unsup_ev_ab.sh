set -e
timeout -k 10 600 python3 -u -m pytest tests -x -q -m gpu -k "unsup or trainer" --timeout 120 --timeout-method thread 2>&1 | tail -2
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 1500 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'), 'loss', d.get('final_loss'))"; }
for r in 1 2; do
echo "products unsup fork   : $(run --workload products --variant unsup)"
echo "products unsup no fork: $(SAGE_NO_UNSUP_FORK=1 run --workload products --variant unsup)"
done
echo "reddit trainer fork   : $(run --variant trainer)"
echo "reddit unsup fork   : $(run --variant unsup)"
echo "products trainer fork   : $(run --workload products --variant trainer)"
