set -e
export TMPDIR=/tmp
R=/tmp/tls_raw; rm -rf $R; mkdir -p $R gpurun_out/tls
for w in ppi products; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$w -o t -- python3 bench.py --workload $w --cpu-steps 0 --no-probe --steps 200 --settle-steps 50 > gpurun_out/tls/$w.json 2> gpurun_out/tls/$w.err
python3 profiles/timeline.py $R/$w/t_kernel_trace.csv 3 > gpurun_out/tls/$w.timeline.txt
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/u -o t -- python3 bench.py --workload products --variant unsup --cpu-steps 0 --no-probe --steps 200 --settle-steps 50 > gpurun_out/tls/u.json 2> gpurun_out/tls/u.err
python3 profiles/timeline.py $R/u/t_kernel_trace.csv 3 > gpurun_out/tls/unsup.timeline.txt
