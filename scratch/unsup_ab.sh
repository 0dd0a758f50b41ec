set -e
mkdir -p gpurun_out/unsup_ab
timeout -k 10 200 python3 -u -m pytest tests/test_gpu_linear.py -x -q -k "dinput2 or fusion_flags" --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 100 python3 scratch/unsup_pieces.py
echo "--- serial d-input pair"
SAGE_DINPUT2_SERIAL=1 timeout -k 10 100 python3 scratch/unsup_pieces.py | grep dinput2
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 400 "$@" | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do
  echo "unsup new  : $(run --workload products --variant unsup)"
  echo "unsup old  : $(SAGE_DINPUT2_SERIAL=1 SAGE_NO_ZEROED_HINT=1 run --workload products --variant unsup)"
  echo "unsup hint-only off : $(SAGE_NO_ZEROED_HINT=1 run --workload products --variant unsup)"
done
echo "trainer new: $(run --variant trainer)"
echo "trainer old: $(SAGE_DINPUT2_SERIAL=1 SAGE_NO_ZEROED_HINT=1 run --variant trainer)"
echo "max_pool new: $(run --aggregator max_pool)"
echo "max_pool old: $(SAGE_NO_ZEROED_HINT=1 run --aggregator max_pool)"
echo "max_pool new: $(run --aggregator max_pool)"
echo "max_pool old: $(SAGE_NO_ZEROED_HINT=1 run --aggregator max_pool)"
echo "headline new: $(run)"
echo "headline old: $(SAGE_NO_ZEROED_HINT=1 run)"
