set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 600 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('schedule','')[:30])"; }
for r in 1 2; do
echo "products sup auto      : $(run --workload products)"
echo "products sup overlapped: $(run --workload products --overlapped)"
done
echo "products gcn auto      : $(run --workload products --aggregator gcn)"
echo "products gcn overlapped: $(run --workload products --aggregator gcn --overlapped)"
echo "products max auto      : $(run --workload products --aggregator max)"
echo "products max overlapped: $(run --workload products --aggregator max --overlapped)"
