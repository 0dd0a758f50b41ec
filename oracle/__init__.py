"""CPU restatement of the reference's hot path — TEST INFRASTRUCTURE ONLY.

Nothing under graphsage_amd/ imports this package.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may use it, and only as the checker / the CPU number
reported beside the GPU number — never as the thing measured or shipped.

Parity status: **parity unpinned at the third-party boundary.**  The reference
(wilcoln/graphsage) delegates every operation on this path to torch_geometric /
torch_sparse / torch_scatter / torch_cluster (un-pinned in its env.yml:6-15, absent from
/root/reference and from this image), ships no tests (graphsage/tests/__init__.py is
empty) and no golden vectors.  So this oracle restates (i) the reference's own Python
(layers/sage.py, models/graphsage.py, samplers/uniform_loader.py, samplers/utils.py,
trainers/*, experiments/fig2a/*) line by line and (ii) the published semantics of the
upstream ops at the call sites cited in each function.  It is pinned by hand-computed
known-answer tests (tests/test_oracle_kat.py) and by the Random123 known-answer vectors for
Philox4x32-10, not by outputs of the reference itself.
"""
