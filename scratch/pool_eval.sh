# pooling evaluation pass: relu(pool(x)) once over all nodes (default) against per batch (SAGE_NO_EVAL_POOL_ALL=1)
set -e
for v in "" 1; do
  SAGE_NO_EVAL_POOL_ALL=$v timeout -k 10 400 python3 -m graphsage_amd.experiments.fig2a --dataset reddit-shaped --aggregators max_pool mean_pool --epochs 2 2>/dev/null | sed "s/^/NO_EVAL_POOL_ALL=$v /"
done
