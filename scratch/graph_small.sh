set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 1000 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d.get('schedule','')[:20])"; }
echo "ppi eager      : $(run --workload ppi)"
echo "ppi graph      : $(run --workload ppi --graph)"
echo "products eager : $(run --workload products)"
echo "products graph : $(run --workload products --graph)"
echo "reddit eager   : $(run)"
echo "reddit graph   : $(run --graph)"
