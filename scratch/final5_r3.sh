set -e
mkdir -p gpurun_out/final5
timeout -k 10 600 python3 bench.py > gpurun_out/final5/default.json 2> gpurun_out/final5/default.err
timeout -k 10 600 python3 bench.py > gpurun_out/final5/default2.json 2> gpurun_out/final5/default2.err
timeout -k 10 300 python3 bench.py --workload products --cpu-steps 0 --steps 2000 > gpurun_out/final5/products.json 2>/dev/null
timeout -k 10 300 python3 bench.py --aggregator max_pool --cpu-steps 0 > gpurun_out/final5/max_pool.json 2>/dev/null
timeout -k 10 300 python3 bench.py --graph --cpu-steps 0 --steps 500 > gpurun_out/final5/graph.json 2>/dev/null
for f in default default2 products max_pool graph; do python3 -c "
import json,sys,os
p='gpurun_out/final5/$f.json'
d=json.loads(open(p).read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$f', d['ms_per_step'], d['value'], r.get('frac'), r.get('avg_launch_us'), r.get('overlapped_avg_launch_us'), (d.get('schedule') or '')[:30])"; done
