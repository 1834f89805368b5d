"""Graph ingest for the node-classification path (reference node_classification/utils.py:33-146), without networkx:
Planetoid pickles -> raw 0/1 symmetric adjacency (no added self loops, not normalised: quirk Q11), `add_all` neighbour
lists in ascending column order (utils.py:97-100), dense features, labels and the reference's fixed splits."""
import pickle
import sys

import numpy as np
import scipy.sparse as sp
import torch


def parse_index_file(filename):
    return [int(line.strip()) for line in open(filename)]


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


def load_data(dataset, data_dir="data"):
    """utils.py:33-119.  Returns add_all, adj (torch sparse COO), features, labels, idx_train, idx_val, idx_test."""
    names = ['x', 'y', 'tx', 'ty', 'allx', 'ally']
    objects = []
    for name in names:
        with open("{}/ind.{}.{}".format(data_dir, dataset, name), 'rb') as f:
            objects.append(pickle.load(f, encoding='latin1'))
    x, y, tx, ty, allx, ally = tuple(objects)
    test_idx_reorder = parse_index_file("{}/ind.{}.test.index".format(data_dir, dataset))
    test_idx_range = np.sort(test_idx_reorder)
    if dataset == 'citeseer':       # isolated test nodes become zero rows (utils.py:56-65)
        full = list(range(min(test_idx_reorder), max(test_idx_reorder) + 1))
        tx_extended = sp.lil_matrix((len(full), x.shape[1]))
        tx_extended[test_idx_range - min(test_idx_range), :] = tx
        tx = tx_extended
        ty_extended = np.zeros((len(full), y.shape[1]))
        ty_extended[test_idx_range - min(test_idx_range), :] = ty
        ty = ty_extended
    features = sp.vstack((allx, tx)).tolil()
    features[test_idx_reorder, :] = features[test_idx_range, :]
    adj, add_all = load_graph(dataset, data_dir)
    labels = np.vstack((ally, ty))
    labels[test_idx_reorder, :] = labels[test_idx_range, :]
    idx_test = test_idx_range.tolist()
    extra = {'cora': 1068, 'citeseer': 1707, 'pubmed': 18157}[dataset]        # utils.py:80-94
    idx_train = range(len(y) + extra)
    idx_val = range(len(y) + extra, len(y) + extra + 500)
    features = torch.FloatTensor(np.array(features.todense()))
    if dataset == "citeseer":
        labels = torch.LongTensor([int(np.where(l == 1)[0][0]) if l.any() else 0 for l in labels])
    else:
        labels = torch.LongTensor(np.where(labels)[1])
    return (add_all, sparse_mx_to_torch_sparse_tensor(adj), features, labels, torch.LongTensor(idx_train),
            torch.LongTensor(idx_val), torch.LongTensor(idx_test))


def normalize(mx):
    rowsum = np.array(mx.sum(1))
    r_inv = np.power(rowsum, -1.0).flatten()
    r_inv[np.isinf(r_inv)] = 0.
    return sp.diags(r_inv).dot(mx)


def accuracy(output, labels):
    preds = output.max(1)[1].type_as(labels)
    return preds.eq(labels).double().sum() / len(labels)


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    m = sparse_mx.tocoo().astype(np.float32)
    indices = torch.from_numpy(np.vstack((m.row, m.col)).astype(np.int64))
    return torch.sparse_coo_tensor(indices, torch.from_numpy(m.data), torch.Size(m.shape))
