set -e
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 600 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('host_issue_ms_per_step'), d.get('schedule','')[:40])"; }
U="--workload products --variant unsup"
echo "overlapped: $(run $U)"
echo "serial    : $(run $U --serial)"
echo "overlapped: $(run $U)"
echo "serial    : $(run $U --serial)"
