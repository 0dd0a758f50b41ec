"""to_csc restated (torch_geometric.loader.utils.to_csc, PyG 2.0.x, called at reference
samplers/uniform_loader.py:32):

    perm = (col * N + row).argsort();  row = row[perm];  colptr = ind2ptr(col[perm], N)

The stable argsort makes `perm` unique when (row, col) pairs repeat.
"""
import numpy as np


def to_csc(edge_index, num_nodes):
    edge_index = np.asarray(edge_index, dtype=np.int64)
    row, col = edge_index[0], edge_index[1]
    perm = np.argsort(col * np.int64(num_nodes) + row, kind="stable")
    row_sorted = row[perm]
    counts = np.bincount(col, minlength=num_nodes).astype(np.int64)
    colptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=colptr[1:])
    return colptr, row_sorted, perm.astype(np.int64)
