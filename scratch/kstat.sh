# kernel durations of the serial headline step (short trace): bash scratch/kstat.sh
set -e
export TMPDIR=/tmp
rm -rf /tmp/ks && mkdir -p /tmp/ks gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 bench.py --cpu-steps 0 --no-variant --no-probe --serial --steps 200 --settle-steps 100 > gpurun_out/ks.json 2> gpurun_out/ks.err
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/ks/ks_kernel_stats.csv')))
for r in rows[:12]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
