"""Graph inputs for the layer: synthetic CSR graphs of stated |V| / |E|, the GCN
symmetric normalisation of the reference (sym_norm2, SG.py:18-51) and loaders for the
reference's CSR-text matrix format (main_float.cpp:415-659).

Generation runs on the GPU with torch (plumbing: random numbers, sort, unique); the result
is handed to the HIP path as int32 rowptr / colidx and fp16|fp32 values.
"""
import re

import numpy as np
import torch

from .ops import Csr


def _finish(row, col, n, dtype, self_loops, normalize):
    """coalesce -> (optional) self loops -> sort by (row, col) -> CSR with sym-norm values."""
    if self_loops:
        loops = torch.arange(n, device=row.device, dtype=torch.int64)
        row = torch.cat([row, loops])
        col = torch.cat([col, loops])
    key = torch.unique(row * n + col)                 # sorted, duplicates removed
    row = torch.div(key, n, rounding_mode="floor")
    col = key - row * n
    del key
    counts = torch.bincount(row, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=row.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    if normalize:
        # sym_norm2 with unit edge weights: deg = row sums (self loop included),
        # value = deg^-1/2[row] * deg^-1/2[col]   (SG.py:46-51)
        dis = counts.to(torch.float32).pow(-0.5)
        dis[torch.isinf(dis)] = 0
        val = dis[row] * dis[col]
    else:
        val = torch.ones(row.numel(), dtype=torch.float32, device=row.device)
    return Csr(rowptr.to(torch.int32), col.to(torch.int32), val.to(dtype), n)


def uniform_graph(n, n_edges, seed=12345, device="cuda", dtype=torch.float16, self_loops=True, normalize=True):
    """row, col ~ U[0, n) -- SURVEY 8d generator (u)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.randint(0, n, (n_edges,), generator=g, device=device, dtype=torch.int64)
    col = torch.randint(0, n, (n_edges,), generator=g, device=device, dtype=torch.int64)
    return _finish(row, col, n, dtype, self_loops, normalize)


def rmat_graph(scale, n_edges, a=0.57, b=0.19, c=0.19, seed=12345, device="cuda", dtype=torch.float16,
               self_loops=True, normalize=True):
    """R-MAT with quadrant probabilities a/b/c/(1-a-b-c) -- SURVEY 8d generator (p); n = 2**scale."""
    n = 1 << scale
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.zeros(n_edges, dtype=torch.int64, device=device)
    col = torch.zeros(n_edges, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(n_edges, generator=g, device=device)
        row_bit = (u >= a + b).to(torch.int64)                       # quadrants c, d
        col_bit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)   # quadrants b, d
        row = row * 2 + row_bit
        col = col * 2 + col_bit
    return _finish(row, col, n, dtype, self_loops, normalize)


def rmat_graph_n(n, n_edges, a=0.57, b=0.19, c=0.19, seed=12345, device="cuda", dtype=torch.float16, self_loops=True,
                 normalize=True):
    """R-MAT on any number of nodes (the Reddit / ogbn-products / ogbn-arxiv shapes are not powers of two): generated
    on 2^ceil(log2 n) ids, every id folded onto [0, n) by id * n >> scale, which keeps neighbouring ids -- and with
    them the skew of the degree distribution -- together."""
    scale = max(1, (n - 1).bit_length())
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.zeros(n_edges, dtype=torch.int64, device=device)
    col = torch.zeros(n_edges, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(n_edges, generator=g, device=device)
        row = row * 2 + (u >= a + b).to(torch.int64)
        col = col * 2 + (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
    row = (row * n) >> scale
    col = (col * n) >> scale
    return _finish(row, col, n, dtype, self_loops, normalize)


def block_local_graph(n, n_edges, n_blocks, p_local=0.9, seed=12345, device="cuda", dtype=torch.float16):
    """Uniform graph whose edges stay inside the row's block (of n/n_blocks nodes) with
    probability p_local: the partition-friendly stand-in used for the multi-GPU halo path."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.randint(0, n, (n_edges,), generator=g, device=device, dtype=torch.int64)
    blk = n // n_blocks
    local = torch.rand(n_edges, generator=g, device=device) < p_local
    off = torch.randint(0, blk, (n_edges,), generator=g, device=device, dtype=torch.int64)
    anywhere = torch.randint(0, n, (n_edges,), generator=g, device=device, dtype=torch.int64)
    base = torch.clamp(torch.div(row, blk, rounding_mode="floor"), max=n_blocks - 1) * blk
    col = torch.where(local, base + off, anywhere)
    return _finish(row, col, n, dtype, True, True)


# ---- the reference's CSR-text matrices (three lines: rowptr / colidx / values) ----------------
def _tokens(line):
    return [t for t in re.split(r"[,\s]+", line.strip()) if t]


def read_csr_text(path):
    """main_float.cpp:415-536 format; values parsed text -> float32 as `ss >> float` does."""
    with open(path) as f:
        lines = [ln for ln in f.read().split("\n") if ln.strip()]
    rowptr = np.array(_tokens(lines[0]), dtype=np.int64).astype(np.int32)
    col = np.array(_tokens(lines[1]), dtype=np.int64).astype(np.int32)
    val = np.array([float(t) for t in _tokens(lines[2])], dtype=np.float64).astype(np.float32)
    return rowptr, col, val


def read_weights_text(path, n_rows=None, n_cols=None):
    """M_fea lines x P comma separated floats, row-major [M_fea][P] (main_float.cpp:149-200)."""
    rows = []
    with open(path) as f:
        for ln in f:
            if ln.strip():
                rows.append([float(t) for t in _tokens(ln)])
    w = np.array(rows, dtype=np.float64).astype(np.float32)
    return w[:n_rows, :n_cols]


def csr_from_numpy(rowptr, col, val, n_cols, dtype=torch.float16, device="cuda"):
    return Csr(torch.as_tensor(np.asarray(rowptr, np.int32), device=device),
               torch.as_tensor(np.asarray(col, np.int32), device=device),
               torch.as_tensor(np.asarray(val, np.float32), device=device).to(dtype), n_cols)
