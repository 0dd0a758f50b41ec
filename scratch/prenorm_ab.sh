# same-box A/B: hidden layer's normalise pass folded into the class-logit kernel (default) or kept (SAGE_NO_PRENORM=1)
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 600 "$@" > gpurun_out/pn.txt 2>gpurun_out/pn.err || { tail -5 gpurun_out/pn.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/pn.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"], (d["roofline"] or {}).get("serial_ms_per_step"))
PY
}
for r in 1 2 3; do
LABEL=prenorm run
LABEL=normpass SAGE_NO_PRENORM=1 run
done
