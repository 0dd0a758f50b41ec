set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 800 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for a in max_pool mean_pool; do
for r in 1 2; do
echo "$a d-input fp32    : $(run --aggregator $a)"
echo "$a forward-type x3 : $(SAGE_POOL_DAGG_X3=1 run --aggregator $a)"
done; done
