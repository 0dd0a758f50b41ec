set -e
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_engine.py -x -q -k "unsup" --timeout 120 --timeout-method thread 2>&1 | tail -2
run() { timeout -k 10 200 python3 bench.py --cpu-steps 0 --no-probe --steps 400 "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
U="--workload products --variant unsup"
echo "unsup base            : $(run $U)"
echo "unsup fwd 1 step/split: $(SAGE_FWD_SPLIT_STEPS=1 run $U)"
for s in 128 256 384; do echo "unsup dw slots $s     : $(SAGE_DW_SPLIT_SLOTS=$s run $U)"; done
echo "unsup base            : $(run $U)"
for s in 64 128 256 384 512; do echo "pieces dw slots $s: $(SAGE_DW_SPLIT_SLOTS=$s timeout -k 10 100 python3 scratch/unsup_pieces.py 2>/dev/null | grep 'dweight2 rows random')"; done
echo "pieces fwd: $(timeout -k 10 100 python3 scratch/unsup_pieces.py 2>/dev/null | grep 'layer-1 fwd')"
echo "pieces fwd 1 step: $(SAGE_FWD_SPLIT_STEPS=1 timeout -k 10 100 python3 scratch/unsup_pieces.py 2>/dev/null | grep 'layer-1 fwd')"
