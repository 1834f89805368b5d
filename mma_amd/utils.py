"""Graph ingest for the node-classification path (reference node_classification/utils.py:33-146), without networkx:
Planetoid pickles -> raw 0/1 symmetric adjacency (no added self loops, not normalised: quirk Q11), `add_all` neighbour
lists in ascending column order (utils.py:97-100), dense features, labels and the reference's fixed splits."""
import pickle
import sys

import numpy as np
import scipy.sparse as sp
import torch


def parse_index_file(filename):
    with open(filename) as f:
        return [int(tok) for tok in f.read().split()]


def adjacency_from_dict_of_lists(graph, n=None):
    """= nx.adjacency_matrix(nx.from_dict_of_lists(graph)) (utils.py:71): symmetric, duplicates merged, entries 1."""
    n = len(graph) if n is None else n
    src = np.concatenate([np.full(len(v), k, dtype=np.int64) for k, v in graph.items()]) if n else np.zeros(0, np.int64)
    dst = np.concatenate([np.asarray(v, dtype=np.int64) for v in graph.values()]) if n else np.zeros(0, np.int64)
    a = sp.coo_matrix((np.ones(2 * len(src), np.float32), (np.concatenate([src, dst]), np.concatenate([dst, src]))),
                      shape=(n, n)).tocsr()
    a.data[:] = 1.0
    a.sort_indices()
    return a


def load_graph(dataset, data_dir="data"):
    """Structure only: (adj csr_matrix, add_all list of int arrays)."""
    with open("{}/ind.{}.graph".format(data_dir, dataset), "rb") as f:
        graph = pickle.load(f, encoding="latin1") if sys.version_info > (3, 0) else pickle.load(f)
    adj = adjacency_from_dict_of_lists(graph)
    return adj, [adj.indices[adj.indptr[i]:adj.indptr[i + 1]] for i in range(adj.shape[0])]


_TRAIN_EXTRA = {'cora': 1068, 'citeseer': 1707, 'pubmed': 18157}      # utils.py:80-94: train = labelled + this many more


def _unpickle(path):
    with open(path, 'rb') as f:
        return pickle.load(f, encoding='latin1')


def load_data(dataset, data_dir="data"):
    """The reference's Planetoid loader (utils.py:33-119) -> add_all, adj (torch sparse COO), features, labels,
    idx_train, idx_val, idx_test, with the same fixed splits."""
    part = {n: _unpickle("{}/ind.{}.{}".format(data_dir, dataset, n)) for n in ('x', 'y', 'tx', 'ty', 'allx', 'ally')}
    test_order = np.asarray(parse_index_file("{}/ind.{}.test.index".format(data_dir, dataset)))
    test_sorted = np.sort(test_order)
    tx, ty = part['tx'], part['ty']
    if dataset == 'citeseer':       # test ids with no node: zero feature / label rows keep the index range dense
        lo, span = test_order.min(), test_order.max() - test_order.min() + 1
        tx_full = sp.lil_matrix((span, part['x'].shape[1]))
        tx_full[test_sorted - lo, :] = tx
        ty_full = np.zeros((span, part['y'].shape[1]))
        ty_full[test_sorted - lo, :] = ty
        tx, ty = tx_full, ty_full
    feats = sp.vstack((part['allx'], tx)).tolil()
    feats[test_order, :] = feats[test_sorted, :]
    onehot = np.vstack((part['ally'], ty))
    onehot[test_order, :] = onehot[test_sorted, :]
    adj, add_all = load_graph(dataset, data_dir)
    n_lab = len(part['y'])
    n_train = n_lab + _TRAIN_EXTRA[dataset]
    if dataset == "citeseer":       # rows without a label become class 0
        cls = np.where(onehot.any(1), onehot.argmax(1), 0)
    else:
        cls = np.where(onehot)[1]
    return (add_all, sparse_mx_to_torch_sparse_tensor(adj), torch.FloatTensor(np.asarray(feats.todense())),
            torch.LongTensor(cls), torch.arange(n_train), torch.arange(n_train, n_train + 500),
            torch.LongTensor(test_sorted.tolist()))


def normalize(mx):
    """Row-normalise (the reference defines it, utils.py:122-129, but never calls it)."""
    with np.errstate(divide='ignore'):
        inv = 1.0 / np.asarray(mx.sum(1), dtype=np.float64).ravel()
    inv[~np.isfinite(inv)] = 0.0
    return sp.diags(inv) @ mx


def accuracy(output, labels):
    return (output.argmax(1) == labels).double().mean()


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    coo = sparse_mx.tocoo()
    idx = torch.from_numpy(np.stack([coo.row, coo.col]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data.astype(np.float32)), tuple(coo.shape))
