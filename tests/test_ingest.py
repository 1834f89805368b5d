"""Graph ingest: the scipy-only loader must reproduce the reference's networkx-built adjacency.  The
golden fixtures hold the CSR the reference produced for Cora and Pubmed; the pickles themselves live under
/root/reference (build container only), so the comparison is skipped where they are absent."""
import os

import numpy as np
import pytest

from golden_util import Golden

DATA = "/root/reference/node_classification/data"


@pytest.mark.parametrize("dataset,gold", [("cora", "cora_h64"), ("pubmed", "pubmed_h16")])
def test_loader_reproduces_reference_graph(dataset, gold):
    if not os.path.exists(os.path.join(DATA, "ind.%s.graph" % dataset)):
        pytest.skip("Planetoid pickles not available here")
    from mma_amd.utils import load_graph
    g = Golden(gold)
    adj, add_all = load_graph(dataset, DATA)
    assert adj.shape[0] == g.N and np.array_equal(adj.indptr, g.rowptr) and np.array_equal(adj.indices, g.col)
    assert (adj.data == 1).all() and (adj != adj.T).nnz == 0
    assert all(np.array_equal(a, b) for a, b in zip(add_all[:50], g.add_all[:50]))


def test_load_data_cora_shapes():
    if not os.path.exists(os.path.join(DATA, "ind.cora.allx")):
        pytest.skip("Planetoid pickles not available here")
    from mma_amd.utils import load_data
    add_all, adj, features, labels, idx_train, idx_val, idx_test = load_data("cora", DATA)
    assert features.shape == (2708, 1433) and labels.shape == (2708,) and int(labels.max()) == 6
    assert len(add_all) == 2708 and adj.shape == (2708, 2708) and adj._nnz() == 10556
    assert len(idx_train) == 140 + 1068 and len(idx_val) == 500 and len(idx_test) == 1000
