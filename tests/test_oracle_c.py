"""The C restatement of the sampler agrees bit for bit with the Python restatement."""
import numpy as np
import pytest

from oracle import csc, sampler, sampler_c
from tests.helpers import random_graph


@pytest.mark.skipif(not sampler_c.available(), reason="oracle/_build not built (run __graft_entry__.build())")
@pytest.mark.parametrize("fanout", [[5, 3], [25, 10], [-1, 2], [70, 4], [3, 3, 3]])
def test_c_sampler_matches_python(fanout):
    ei = random_graph(800, 40000, seed=5, isolated=6, hubs=3)
    colptr, row, _ = csc.to_csc(ei, 800)
    seeds = np.random.default_rng(1).choice(800, 40, replace=False)
    seeds[3] = seeds[2]                      # a duplicate seed
    seeds[0] = 0                             # an isolated seed
    a = sampler.neighbor_sample(colptr, row, seeds, fanout, 9, 123456789012345)
    b = sampler_c.neighbor_sample(colptr, row, seeds, fanout, 9, 123456789012345)
    for x, y in zip(a[:4], b[:4]):
        assert x.tolist() == y.tolist()
    assert a[4] == b[4] and a[5] == b[5]
