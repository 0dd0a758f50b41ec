set -e
run() { timeout -k 10 60 python3 bench.py --cpu-steps 0 --no-probe --steps 1500 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"; }
for r in 1 2 3; do
echo "gate at step start : $(run)"
echo "gate at mid step   : $(SAGE_MID_GATE=1 run)"
done
