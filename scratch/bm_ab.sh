# same-box A/B of the 128-row tile of the bf16x3 whole-tile forward (SAGE_GEMM_BM=64: 64-row tile)
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 300 "$@" > gpurun_out/bab.txt 2>gpurun_out/bab.err || { tail -5 gpurun_out/bab.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/bab.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"], d["roofline"]["achieved"])
PY
}
LABEL=pool_bm128 run --aggregator max_pool
LABEL=pool_bm64 SAGE_GEMM_BM=64 run --aggregator max_pool
LABEL=pool_bm128 run --aggregator max_pool
