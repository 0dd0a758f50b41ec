set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 800 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"; }
for a in max_pool mean_pool; do
for r in 1 2; do
echo "$a pool d-input fp32    : $(run --aggregator $a)"
echo "$a pool d-input fwd x3  : $(SAGE_POOL_DIN_X3=1 run --aggregator $a)"
done; done
echo "max_pool all fp32 d-inputs: $(SAGE_NO_DAGG_X3=1 run --aggregator max_pool)"
