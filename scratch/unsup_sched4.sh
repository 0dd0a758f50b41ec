set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 600 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('schedule','')[:30])"; }
for r in 1 2; do
echo "reddit trainer overlapped: $(run --variant trainer)"
echo "reddit trainer serial    : $(run --variant trainer --serial)"
done
echo "reddit unsup overlapped: $(run --variant unsup)"
echo "reddit unsup serial    : $(run --variant unsup --serial)"
echo "products trainer overlapped: $(run --workload products --variant trainer --overlapped)"
echo "products trainer serial    : $(run --workload products --variant trainer)"
