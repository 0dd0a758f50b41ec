set -e
mkdir -p gpurun_out/final3
timeout -k 10 600 python3 bench.py > gpurun_out/final3/default.json 2> gpurun_out/final3/default.err
timeout -k 10 300 python3 bench.py --workload ppi --cpu-steps 0 --steps 2000 > gpurun_out/final3/ppi.json 2>/dev/null
timeout -k 10 300 python3 bench.py --workload products --variant unsup --cpu-steps 0 --steps 2000 > gpurun_out/final3/products_unsup.json 2>/dev/null
timeout -k 10 300 python3 bench.py --workload products --cpu-steps 0 --steps 2000 > gpurun_out/final3/products.json 2>/dev/null
timeout -k 10 300 python3 bench.py --variant trainer --cpu-steps 0 --steps 2000 > gpurun_out/final3/trainer.json 2>/dev/null
for f in default ppi products_unsup products trainer; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/final3/$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d['value'], (d.get('roofline') or {}).get('frac'))"; done
