# same-box A/B of the XCD-aware tile mapping (SAGE_GEMM_XCD=0: plain order)
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 400 "$@" > gpurun_out/xab.txt 2>gpurun_out/xab.err || { tail -5 gpurun_out/xab.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/xab.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"])
PY
}
LABEL=pool_xcd run --aggregator max_pool
LABEL=pool_plain SAGE_GEMM_XCD=0 run --aggregator max_pool
LABEL=pool_xcd run --aggregator max_pool
LABEL=mean_xcd run
LABEL=mean_plain SAGE_GEMM_XCD=0 run
LABEL=mean_xcd run
