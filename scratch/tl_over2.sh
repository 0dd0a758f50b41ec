set -e
export TMPDIR=/tmp
R=/tmp/tlo_raw; rm -rf $R; mkdir -p $R gpurun_out/tlo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/o -o t -- python3 bench.py --cpu-steps 0 --no-probe --no-variant --steps 300 --settle-steps 100 > gpurun_out/tlo/o.json 2> gpurun_out/tlo/o.err
python3 - <<'PY' > gpurun_out/tlo/overlapped.timeline.txt
import csv
rows=list(csv.DictReader(open('/tmp/tlo_raw/o/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam' in r['Kernel_Name']]
a,b=idx[-4],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('sage::(anonymous namespace)::','').replace('void ','')[:46]
    print(f"{(s-t0)/1e3:8.1f} -> {(e-t0)/1e3:8.1f}  dur={(e-s)/1e3:6.1f}  q={r.get('Queue_Id','?'):>3s}  {name}")
PY
