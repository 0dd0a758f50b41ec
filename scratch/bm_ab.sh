# same-box A/B of the 128-row tiles of the bf16x3 whole-tile forward / weight gradients (SAGE_GEMM_BM=64: 64-row tiles)
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 600 "$@" > gpurun_out/bab.txt 2>gpurun_out/bab.err || { tail -5 gpurun_out/bab.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/bab.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"], d["roofline"]["achieved"], (d["roofline"] or {}).get("serial_ms_per_step"))
PY
}
LABEL=mean_bm128 run
LABEL=mean_bm64 SAGE_GEMM_BM=64 run
LABEL=mean_bm128 run
LABEL=mean_bm64 SAGE_GEMM_BM=64 run
LABEL=pool_bm128 run --aggregator max_pool
LABEL=pool_bm64 SAGE_GEMM_BM=64 run --aggregator max_pool
LABEL=meanpool_bm128 run --aggregator mean_pool
