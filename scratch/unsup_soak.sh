set -e
run() { timeout -k 10 300 python3 bench.py --cpu-steps 0 --no-probe "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'final_loss', d.get('final_loss'))"; }
for st in 100 1500; do
echo "trainer fork    $st: $(run --variant trainer --steps $st)"
echo "trainer no fork $st: $(SAGE_NO_UNSUP_FORK=1 run --variant trainer --steps $st)"
echo "products unsup fork    $st: $(run --workload products --variant unsup --steps $st)"
echo "products unsup no fork $st: $(SAGE_NO_UNSUP_FORK=1 run --workload products --variant unsup --steps $st)"
done
echo "trainer fork 5000: $(run --variant trainer --steps 5000)"
