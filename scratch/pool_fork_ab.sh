# same-box A/B of the pool step's forked layer-0 weight gradients (SAGE_NO_POOL_FORK=1: one stream)
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --no-probe --steps 300 "$@" > gpurun_out/pf.txt 2>gpurun_out/pf.err || { tail -5 gpurun_out/pf.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/pf.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"])
PY
}
for agg in max_pool mean_pool; do
LABEL=${agg}_fork run --aggregator $agg
LABEL=${agg}_one SAGE_NO_POOL_FORK=1 run --aggregator $agg
LABEL=${agg}_fork run --aggregator $agg
LABEL=${agg}_one SAGE_NO_POOL_FORK=1 run --aggregator $agg
done
