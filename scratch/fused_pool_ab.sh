# same-box A/B: the pooling aggregators' last layer in one launch (sage_logit_layer_pool_fwd_bwd, default) against the separate kernels (SAGE_NO_FUSED_POOL_LOGIT=1)
set -e
for rep in 1 2; do
  for v in "" 1; do
    for agg in max_pool mean_pool; do
      SAGE_NO_FUSED_POOL_LOGIT=$v timeout -k 10 200 python bench.py --aggregator $agg --cpu-steps 0 --no-variant --no-probe --steps 300 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$agg NO_FUSED_POOL_LOGIT=%s' % '$v', d['ms_per_step'])"
    done
  done
done
