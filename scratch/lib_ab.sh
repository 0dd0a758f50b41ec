# same-box A/B of library builds (SAGE_HIP_LIB): old, a = packed residual subtraction only, b = + uniform branches without a memory clobber, new = in-tree
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 600 "$@" > gpurun_out/vab.txt 2>gpurun_out/vab.err || { tail -5 gpurun_out/vab.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/vab.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"], (d["roofline"] or {}).get("serial_ms_per_step"))
PY
}
for r in 1 2; do
for v in old a b; do LABEL=mean_$v SAGE_HIP_LIB=$PWD/scratch/libsage_$v.so run; done
LABEL=mean_new run
done
for v in old a b; do LABEL=pool_$v SAGE_HIP_LIB=$PWD/scratch/libsage_$v.so run --aggregator max_pool; done
