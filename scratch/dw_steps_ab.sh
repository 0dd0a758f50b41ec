set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 1500 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for st in 4 2 1; do
echo "min steps $st: ppi $(SAGE_DW_MIN_STEPS=$st run --workload ppi)  products unsup $(SAGE_DW_MIN_STEPS=$st run --workload products --variant unsup)  pieces $(SAGE_DW_MIN_STEPS=$st timeout -k 10 100 python3 scratch/unsup_pieces.py 2>/dev/null | grep 'layer-1 dweight2')"
done
echo "min steps 4: ppi $(SAGE_DW_MIN_STEPS=4 run --workload ppi)"
