"""The few torch_geometric pieces the reference's notebooks use around the hot path, restated
with plain torch so the end-to-end runs do not need PyG (it is not installed here): the TU
dataset reader for MUTAG (raw files -> graphs), DataLoader-style batching into one block-
diagonal graph, `to_dense_adj`, `global_mean_pool` (MOL cells 4-10, 18).
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import torch


@dataclass
class Graph:
    x: torch.Tensor            # [n, F] one-hot node labels
    edge_index: torch.Tensor   # [2, E] int64
    y: torch.Tensor            # [1] int64

    @property
    def num_nodes(self):
        return self.x.shape[0]


@dataclass
class Batch:
    x: torch.Tensor
    edge_index: torch.Tensor
    y: torch.Tensor
    batch: torch.Tensor        # [n] graph id of every node
    num_graphs: int

    @property
    def num_nodes(self):
        return self.x.shape[0]

    def to(self, device):
        return Batch(self.x.to(device), self.edge_index.to(device), self.y.to(device), self.batch.to(device),
                     self.num_graphs)


def load_tu_raw(A, graph_indicator, graph_labels, node_labels) -> List[Graph]:
    """TU format (MUTAG/raw/*.txt): A = 1-based (row, col) pairs, graph_indicator = 1-based graph
    id per node, labels {-1, 1} -> {0, 1} as torch_geometric's TUDataset does, node labels one-hot."""
    A = np.asarray(A, np.int64) - 1
    gi = np.asarray(graph_indicator, np.int64) - 1
    nl = np.asarray(node_labels, np.int64)
    labels = np.asarray(graph_labels, np.int64)
    uniq = np.unique(labels)
    y = np.searchsorted(uniq, labels)
    n_feat = int(nl.max()) + 1
    n_graphs = int(gi.max()) + 1
    first = np.searchsorted(gi, np.arange(n_graphs))
    count = np.bincount(gi, minlength=n_graphs)
    edge_graph = gi[A[:, 0]]
    graphs = []
    for g in range(n_graphs):
        e = A[edge_graph == g] - first[g]
        x = torch.zeros((count[g], n_feat))
        x[torch.arange(count[g]), torch.as_tensor(nl[first[g]:first[g] + count[g]])] = 1.0
        graphs.append(Graph(x, torch.as_tensor(e.T.copy()), torch.tensor([y[g]])))
    return graphs


def collate(graphs: List[Graph]) -> Batch:
    xs, es, ys, bs = [], [], [], []
    off = 0
    for i, g in enumerate(graphs):
        xs.append(g.x)
        es.append(g.edge_index + off)
        ys.append(g.y)
        bs.append(torch.full((g.num_nodes,), i, dtype=torch.int64))
        off += g.num_nodes
    return Batch(torch.cat(xs), torch.cat(es, dim=1), torch.cat(ys), torch.cat(bs), len(graphs))


class DataLoader:
    def __init__(self, dataset, batch_size=1, shuffle=False, generator=None):
        self.dataset, self.batch_size, self.shuffle, self.generator = list(dataset), batch_size, shuffle, generator

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            yield collate([self.dataset[j] for j in order[i:i + self.batch_size]])

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size


def to_dense_adj(edge_index, num_nodes=None):
    """[1, N, N] 0/1 adjacency (duplicates add up, as torch_geometric.utils.to_dense_adj)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    adj = torch.zeros((n, n), dtype=torch.float32, device=edge_index.device)
    adj.index_put_((edge_index[0], edge_index[1]), torch.ones(edge_index.shape[1], device=edge_index.device),
                   accumulate=True)
    return adj.unsqueeze(0)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    out = torch.zeros((size, x.shape[1]), dtype=x.dtype, device=x.device)
    out.index_add_(0, batch, x)
    cnt = torch.bincount(batch, minlength=size).clamp(min=1).to(x.dtype)
    return out / cnt.unsqueeze(1)


def add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes):
    """Self loop (weight fill_value) for every node that has none; existing loops keep their weight."""
    row, col = edge_index
    has = torch.zeros(num_nodes, dtype=torch.bool, device=row.device)
    has[row[row == col]] = True
    missing = torch.nonzero(~has).reshape(-1)
    ei = torch.cat([edge_index, torch.stack([missing, missing])], dim=1)
    ew = torch.cat([edge_weight, torch.full((missing.numel(),), float(fill_value), dtype=edge_weight.dtype,
                                            device=row.device)])
    return ei, ew


def sort_edge_index(edge_index, edge_weight, num_nodes):
    key = edge_index[0] * num_nodes + edge_index[1]
    perm = torch.argsort(key, stable=True)
    return edge_index[:, perm], edge_weight[perm]
