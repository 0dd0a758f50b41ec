# longer runs on the last tree, the three-forward steps and the narrow-row workloads included:
# the loss stays finite, nothing hangs
set -e
go() { label="$1"; shift; timeout -k 10 300 python bench.py --cpu-steps 0 --no-variant --no-probe --warmup 10 --settle-steps 20 "$@" 2>/dev/null | python3 -c "import sys,json,math; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); assert math.isfinite(d['final_loss']); print('$label', d['steps'], 'steps', d['ms_per_step'], 'ms/step, final loss', d['final_loss'])"; }
go "reddit mean" --steps 50000
go "reddit sum" --aggregator sum --steps 20000
go "reddit max" --aggregator max --steps 20000
go "reddit gcn" --aggregator gcn --steps 20000
go "reddit max_pool" --aggregator max_pool --steps 5000
go "reddit mean_pool" --aggregator mean_pool --steps 5000
go "reddit lstm" --aggregator lstm --steps 1000
go "reddit trainer (sup + unsup)" --variant trainer --steps 10000
go "reddit unsup" --variant unsup --steps 10000
go "products unsup" --workload products --variant unsup --steps 20000
go "products mean" --workload products --steps 30000
go "ppi mean" --workload ppi --steps 30000
