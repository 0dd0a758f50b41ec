set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 600 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('schedule','')[:30])"; }
echo "products unsup: $(run --workload products --variant unsup)"
echo "products sup  : $(run --workload products)"
echo "products sup  : $(run --workload products)"
echo "ppi           : $(run --workload ppi)"
echo "reddit        : $(run)"
echo "reddit trainer: $(run --variant trainer)"
