set -e
timeout -k 10 200 python3 -u -m pytest tests/test_gpu_linear.py -x -q -k "dinput2 or fusion_flags" --timeout 120 --timeout-method thread 2>&1 | tail -2
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 400 "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do
  echo "ppi new  : $(run --workload ppi)"
  echo "ppi old  : $(SAGE_DINPUT2_SERIAL=1 SAGE_NO_ZEROED_HINT=1 run --workload ppi)"
  echo "ppi serial pair, hint on: $(SAGE_DINPUT2_SERIAL=1 run --workload ppi)"
done
for w in cora products; do
  echo "$w new  : $(run --workload $w)"
  echo "$w old  : $(SAGE_DINPUT2_SERIAL=1 SAGE_NO_ZEROED_HINT=1 run --workload $w)"
done
echo "products unsup new: $(run --workload products --variant unsup)"
