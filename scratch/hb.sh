set -e
run() { echo "== $*"; timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 800 "$@" > gpurun_out/hb.txt 2>gpurun_out/hb.err || { tail -5 gpurun_out/hb.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/hb.txt") if x.startswith("{")][-1]; d=json.loads(l)
r=d["roofline"] or {}
print(d["ms_per_step"], d["value"], d["host_enqueue_ms_per_step"], r.get("serial_ms_per_step"), r.get("avg_launch_us"), d["final_loss"])
PY
}
run
SAGE_NO_FUSED_LOGIT=1 run
run
