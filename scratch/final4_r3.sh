set -e
mkdir -p gpurun_out/final4
timeout -k 10 300 python3 -u -m pytest tests/test_bench_launch.py tests/test_gpu_dist.py -x -q -m gpu --timeout 200 --timeout-method thread 2>&1 | tail -2
timeout -k 10 600 python3 bench.py > gpurun_out/final4/default.json 2> gpurun_out/final4/default.err
timeout -k 10 300 python3 bench.py --workload ppi --cpu-steps 0 --steps 2000 > gpurun_out/final4/ppi.json 2>/dev/null
timeout -k 10 300 python3 bench.py --workload products --variant unsup --cpu-steps 0 --steps 2000 > gpurun_out/final4/products_unsup.json 2>/dev/null
timeout -k 10 300 python3 bench.py --workload products --cpu-steps 0 --steps 2000 > gpurun_out/final4/products.json 2>/dev/null
timeout -k 10 300 python3 bench.py --variant trainer --cpu-steps 0 --steps 2000 > gpurun_out/final4/trainer.json 2>/dev/null
timeout -k 10 300 python3 bench.py --serial --cpu-steps 0 --steps 1000 > gpurun_out/final4/serial.json 2>/dev/null
timeout -k 10 300 python3 bench.py --force-collective --cpu-steps 0 --steps 1000 > gpurun_out/final4/dp1.json 2>/dev/null || echo "no --force-collective flag"
for f in default ppi products_unsup products trainer serial dp1; do python3 -c "
import json,sys,os
p='gpurun_out/final4/$f.json'
if os.path.exists(p) and os.path.getsize(p):
    d=json.loads(open(p).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print('$f', d['ms_per_step'], d['value'], r.get('frac'), r.get('avg_launch_us'), (d.get('schedule') or '')[:30])"; done
