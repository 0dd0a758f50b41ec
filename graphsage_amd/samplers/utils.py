"""get_pos_neg_batches with the reference's signature (reference graphsage/samplers/utils.py:7-37).

Positive = one uniformly chosen in-neighbour of each batch node in the FULL graph (the
paper's 1-step random-walk co-occurrence), negative = a uniformly random node.  The
reference runs torch_cluster.random_walk on the batch's LOCAL graph with GLOBAL ids as
start nodes (utils.py:22) — an indexing quirk whose result depends on upstream internals and
cannot be reproduced without torch_cluster; parity for this function is therefore unpinned
and the paper's intent is implemented.  Features are gathered on the device
(the reference round-trips through the CPU, utils.py:27,32)."""
from .. import ops
from ..data import Data
from .uniform_loader import device_graph

_counter = [0]


def get_pos_neg_batches(batch, data, seed=0, counter=None):
    """seed / counter: the Philox key and counter of both draws (counter=None: a module-level
    count of calls).  With the batch's own sampler counter this is exactly what the static-shape
    engine draws for that batch (engine.make_pos_neg), and what oracle/sampler.py restates."""
    graph = device_graph(data)
    n_id = batch.n_id.to(graph.device)
    if counter is None:
        _counter[0] += 1
        counter = _counter[0]
    pos_n_id = ops.random_neighbor(graph.colptr, graph.row, n_id, seed, counter)
    neg_n_id = ops.random_nodes(data.num_nodes, n_id.numel(), seed, counter, device=graph.device)
    table = graph.features(data)
    ei = batch.edge_index
    pos_batch = Data(x=ops.gather_rows(table, pos_n_id, checked=True), n_id=pos_n_id, edge_index=ei)
    neg_batch = Data(x=ops.gather_rows(table, neg_n_id, checked=True), n_id=neg_n_id, edge_index=ei)
    return pos_batch, neg_batch
