# same-box A/B: in-degree workspace for every layer (SAGE_POOL_INDEG_ALL=1) or for layer 0 only (default)
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --no-probe --steps 300 "$@" > gpurun_out/pf.txt 2>gpurun_out/pf.err || { tail -5 gpurun_out/pf.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/pf.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"])
PY
}
for agg in max_pool mean_pool; do
LABEL=${agg}_l0only run --aggregator $agg
LABEL=${agg}_all SAGE_POOL_INDEG_ALL=1 run --aggregator $agg
LABEL=${agg}_l0only run --aggregator $agg
LABEL=${agg}_all SAGE_POOL_INDEG_ALL=1 run --aggregator $agg
done
