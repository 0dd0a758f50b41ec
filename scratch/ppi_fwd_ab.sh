set -e
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_linear.py -x -q -k "concat or fusion_flags" --timeout 120 --timeout-method thread 2>&1 | tail -2
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 1500 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; }
for r in 1 2 3; do
echo "ppi whole tiles: $(run --workload ppi)"
echo "ppi stream-K   : $(SAGE_FWD_STREAMK_SHORT=1 run --workload ppi)"
done
