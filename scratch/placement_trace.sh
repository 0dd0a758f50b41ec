# per-kernel durations per engine instance of scratch/placement.py (4 instances; which kernels are slower in a slow placement?)
set -e
export TMPDIR=/tmp
rm -rf /tmp/pl && mkdir -p /tmp/pl gpurun_out
SAGE_PLACEMENT_ITERS=4 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -o pl -- python3 scratch/placement.py mean 1 > gpurun_out/pl.txt 2> gpurun_out/pl.err
cat gpurun_out/pl.txt
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/pl/pl_kernel_trace.csv')))
by = collections.defaultdict(list)
for r in rows:
    by[r['Kernel_Name']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for name, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    if len(v) < 400: continue
    v.sort()
    n = len(v) // 4
    print("%-60s" % name.replace('sage::(anonymous namespace)::', '').replace('void ', '')[:60],
          " ".join("%7.1f" % (sum(d for _, d in v[i * n:(i + 1) * n]) / n / 1e3) for i in range(4)))
PY
