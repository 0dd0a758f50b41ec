set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 800 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for a in max_pool mean_pool; do
for st in 4 2 4 2; do echo "$a min steps $st: $(SAGE_DW_MIN_STEPS=$st run --aggregator $a)"; done
done
for st in 4 2; do echo "trainer min steps $st: $(SAGE_DW_MIN_STEPS=$st run --variant trainer)"; done
for st in 4 2; do echo "products min steps $st: $(SAGE_DW_MIN_STEPS=$st run --workload products)"; done
