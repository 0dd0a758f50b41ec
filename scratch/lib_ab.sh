# same-box A/B of two library builds: scratch/libsage_old.so (SAGE_HIP_LIB) against the in-tree one
set -e
run() { timeout -k 10 200 python bench.py --cpu-steps 0 --no-variant --steps 600 "$@" > gpurun_out/vab.txt 2>gpurun_out/vab.err || { tail -5 gpurun_out/vab.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/vab.txt") if x.startswith("{")][-1]; d=json.loads(l)
print("$LABEL", d["ms_per_step"], d["value"], d["final_loss"], (d["roofline"] or {}).get("serial_ms_per_step"))
PY
}
for r in 1 2; do
LABEL=mean_old SAGE_HIP_LIB=$PWD/scratch/libsage_old.so run
LABEL=mean_new run
done
LABEL=pool_old SAGE_HIP_LIB=$PWD/scratch/libsage_old.so run --aggregator max_pool
LABEL=pool_new run --aggregator max_pool
