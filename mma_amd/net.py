"""Model glue for the graph-regression path (reference graph_regression/mma.py:63-127 `Net`): atom / bond embeddings,
4 x (MMAConv 75->75, towers=5, edge_dim=50  +  BatchNorm), global add pooling, MLP head.  No torch_geometric: BatchNorm is
torch's BatchNorm1d (PyG's BatchNorm wraps exactly that) and global_add_pool is an index_add over the batch vector."""
import torch
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Embedding, Linear, ModuleList, ReLU, Sequential

from .mma_conv import CategoricalEdges, MMAConv


def global_add_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return torch.zeros((size, x.shape[1]), device=x.device, dtype=x.dtype).index_add_(0, batch, x)


class Net(torch.nn.Module):
    categorical_edges = True      # hand the bond types and the embedding table to the layers (same math as the embedded rows)

    def __init__(self, aggregator_list, scaler_list, deg, mask=True, layers=4):
        super().__init__()
        self.node_emb = Embedding(21, 75)
        self.edge_emb = Embedding(4, 50)
        self.convs = ModuleList()
        self.batch_norms = ModuleList()
        for _ in range(layers):                                   # mma.py:91-97 (4 layers, towers=5, hard-coded)
            self.convs.append(MMAConv(in_channels=75, out_channels=75, aggregators=aggregator_list, scalers=scaler_list,
                                      deg=deg, edge_dim=50, towers=5, pre_layers=1, post_layers=1, mask=mask,
                                      divide_input=False))
            self.batch_norms.append(BatchNorm1d(75))
        self.mlp = Sequential(Linear(75, 50), ReLU(), Linear(50, 25), ReLU(), Linear(25, 1))

    def forward(self, x, edge_index, edge_attr, batch):
        x = self.node_emb(x.squeeze())
        if self.categorical_edges and x.is_cuda and edge_attr.dim() == 1:
            edge_attr = CategoricalEdges(edge_attr, self.edge_emb.weight)     # = self.edge_emb(edge_attr), never materialised
        else:
            edge_attr = self.edge_emb(edge_attr)
        for conv, batch_norm in zip(self.convs, self.batch_norms):
            x = F.relu(batch_norm(conv(x, edge_index, edge_attr)))
        x = global_add_pool(x, batch)
        return self.mlp(x)
