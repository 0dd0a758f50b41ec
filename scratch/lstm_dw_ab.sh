set -e
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_linear.py -x -q -k "dweight" --timeout 120 --timeout-method thread 2>&1 | tail -2
run() { timeout -k 10 300 python3 bench.py --cpu-steps 0 --no-probe "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"; }
for r in 1 2; do
echo "lstm new rule : $(run --aggregator lstm --steps 300)"
echo "lstm rows only: $(SAGE_DW_X3_MIN_ROWS=2048 run --aggregator lstm --steps 300)"
done
echo "max_pool new rule : $(run --aggregator max_pool --steps 600)"
echo "max_pool rows only: $(SAGE_DW_X3_MIN_ROWS=2048 run --aggregator max_pool --steps 600)"
echo "headline new rule : $(run --steps 1000)"
echo "headline rows only: $(SAGE_DW_X3_MIN_ROWS=2048 run --steps 1000)"
