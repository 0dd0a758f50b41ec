# where the host's 0.2 ms per step go: cProfile of the headline run (bash scratch/host_prof.sh on the GPU box)
set -e
timeout -k 10 280 python -m cProfile -o /tmp/bench.prof bench.py --cpu-steps 0 --no-variant --no-probe --steps 3000 > gpurun_out/hp.json 2> gpurun_out/hp.err
python - <<'PY' > gpurun_out/host_prof.txt
import pstats
p = pstats.Stats('/tmp/bench.prof')
p.sort_stats('tottime').print_stats(28)
p.sort_stats('cumulative').print_stats(40)
PY
