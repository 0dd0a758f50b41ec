# same-box A/B: the one-wave logit kernel's 64-value reduction on the VALU (v_permlane swaps + DPP, the tree) against the
# ds_bpermute butterfly (scratch/alt/libsage_hip_bpermute.so = the previous commit's library built aside, through SAGE_HIP_LIB;
# not kept in the tree: build it from `git show 78c36da:graphsage_amd/csrc/logit.hip` to run this again)
set -e
for rep in 1 2; do
  for v in "" scratch/alt/libsage_hip_bpermute.so; do
    if [ -n "$v" ]; then export SAGE_HIP_LIB=$PWD/$v; else unset SAGE_HIP_LIB; fi
    timeout -k 10 200 python bench.py --aggregator max --cpu-steps 0 --no-variant --no-probe --steps 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('max lib=%s' % '$v', d['ms_per_step'])"
  done
done
unset SAGE_HIP_LIB
for v in "" scratch/alt/libsage_hip_bpermute.so; do
  if [ -n "$v" ]; then export SAGE_HIP_LIB=$PWD/$v; else unset SAGE_HIP_LIB; fi
  timeout -k 10 300 python3 -m graphsage_amd.experiments.fig2a --dataset reddit-shaped --aggregators mean max --epochs 2 2>/dev/null | cut -c1-200 | sed "s|^|lib=$v |"
done
