# same-box A/B: weight gradients of the three variants of the unsup / trainer step forked (SAGE_UNSUP_FORK=1) or not
# (the switch existed for this measurement only: engine.py read SAGE_UNSUP_FORK when it created its leaf stream; removed after the A/B)
set -e
run() { timeout -k 10 280 python bench.py --cpu-steps 0 --no-variant --no-probe --steps 300 "$@" > gpurun_out/uf.txt 2>gpurun_out/uf.err || { tail -5 gpurun_out/uf.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/uf.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"])
PY
}
for r in 1 2; do
LABEL=trainer_one run --variant trainer
LABEL=trainer_fork SAGE_UNSUP_FORK=1 run --variant trainer
done
for r in 1 2; do
LABEL=products_one run --workload products --variant unsup
LABEL=products_fork SAGE_UNSUP_FORK=1 run --workload products --variant unsup
done
