"""Node-partitioned execution of the layer across the GPUs of one node.

Partition: contiguous 1-D row blocks, exactly the reference's thread split with pointer rebasing
(K.cpp:1378-1382, :3517-3523): rank g owns rows [lo_g, hi_g) of A (CSR slice with its own
rowptr starting at 0 and GLOBAL column indices), the same rows of X, and a replica of W.
Per layer:   H_g = X_g . W   ->   exchange rows of H   ->   D_g = act(A_g . H).

Two exchanges (both are the GPU form of compute1_4 replicating its C block to every ADJ
thread's PIPO, K.cpp:2913-2916, and dsp_kernel_float_adj_4 selecting the block by column,
K.cpp:217-264):
  * "allgather": every rank receives every block of H (one RCCL all_gather_into_tensor);
    A_g keeps global column indices.
  * "halo": every rank receives only the rows its edges reference.  Preprocessing builds, per
    (owner, consumer) pair, the sorted unique list of needed rows and remaps A_g's column indices
    to the compact table [own rows | halo rows grouped by owner]; at run time a gather packs the
    rows each peer asked for and one all_to_all_single moves them.

The collective runs on torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  The local compute is a `Backend`: the HIP library on GPUs; the CPU tests
inject their own callables to exercise the partition / remap / exchange logic without a GPU.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def row_partition(n_rows: int, world: int, rowptr: Optional[torch.Tensor] = None):
    """Boundaries lo[0..world]: equal row counts, or (rowptr given) equal nnz -- the reference
    splits by row count only (K.cpp:3517-3523); nnz balance is what power-law graphs need."""
    if rowptr is None:
        base = n_rows // world
        bounds = [g * base for g in range(world)] + [n_rows]     # remainder to the last, as K.cpp:3522
        return bounds
    nnz = int(rowptr[-1])
    targets = torch.tensor([nnz * g // world for g in range(1, world)], dtype=rowptr.dtype, device=rowptr.device)
    cuts = torch.searchsorted(rowptr.contiguous(), targets).tolist()
    return [0] + [min(int(c), n_rows) for c in cuts] + [n_rows]


def slice_rows(rowptr, col, val, lo, hi):
    """Rows [lo, hi) of a CSR matrix with the row pointer rebased to 0 (reada2, K.cpp:1378-1382)."""
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    return (rowptr[lo:hi + 1] - rowptr[lo]).contiguous(), col[e0:e1].contiguous(), val[e0:e1].contiguous()


def _host_staged(t, group):
    """gloo moves host memory: device tensors are staged through the CPU when the job runs on gloo
    (CPU tests, and rehearsing several ranks on one GPU); RCCL takes device tensors directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_into(out, inp, group=None):
    if _host_staged(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def all_to_all_rows(out, inp, out_splits, in_splits, group=None):
    if _host_staged(inp, group) or (out.is_cuda and dist.get_backend(group) == "gloo"):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


@dataclass
class HaloPlan:
    """Index structures of the halo exchange for one rank (all tensors on the compute device)."""
    bounds: List[int]                 # row partition, len world+1
    rank: int
    col_compact: torch.Tensor         # int32 [nnz_local]: column -> row of the compact table
    send_rows: torch.Tensor           # int64 [sum send_counts]: LOCAL row ids to pack, grouped by consumer
    send_counts: List[int]            # rows sent to each peer
    recv_counts: List[int]            # rows received from each owner
    n_own: int

    @property
    def n_table(self):
        return self.n_own + sum(self.recv_counts)


def build_halo_plan(col_global: torch.Tensor, bounds: List[int], rank: int, group=None) -> HaloPlan:
    """Needs one all_to_all of the request lists (preprocessing, once per graph)."""
    world = len(bounds) - 1
    lo, hi = bounds[rank], bounds[rank + 1]
    dev = col_global.device
    col64 = col_global.to(torch.int64)
    b = torch.tensor(bounds, dtype=torch.int64, device=dev)
    owner = torch.searchsorted(b, col64, right=True) - 1
    need: List[torch.Tensor] = []                    # per owner: sorted unique global rows this rank reads
    for g in range(world):
        need.append(torch.unique(col64[owner == g]) if g != rank else col64.new_empty(0))
    recv_counts = [int(t.numel()) for t in need]
    # compact numbering: own rows first (global - lo), then each owner's halo rows in sorted order
    col_compact = torch.empty_like(col64)
    own = owner == rank
    col_compact[own] = col64[own] - lo
    off = hi - lo
    for g in range(world):
        if g == rank or recv_counts[g] == 0:
            continue
        m = owner == g
        col_compact[m] = off + torch.searchsorted(need[g], col64[m])
        off += recv_counts[g]
    # tell every owner which of its rows we need
    counts_out = torch.tensor(recv_counts, dtype=torch.int64, device=dev)
    counts_in = torch.empty_like(counts_out)
    all_to_all_rows(counts_in, counts_out, None, None, group=group)
    send_counts = [int(c) for c in counts_in.tolist()]
    req_out = torch.cat(need) if sum(recv_counts) else col64.new_empty(0)
    req_in = col64.new_empty(sum(send_counts))
    all_to_all_rows(req_in, req_out, send_counts, recv_counts, group=group)
    return HaloPlan(bounds, rank, col_compact.to(torch.int32), (req_in - lo).contiguous(), send_counts, recv_counts,
                    hi - lo)


@dataclass
class Backend:
    """Local compute: xw(fea_local, Wt) -> H_local [n_local, P];  spmm(adj_csr, table, relu) -> D_local;
    for the overlapped exchange also spmm_partial(adj, table) -> fp32 sums and
    spmm_finish(adj, table, partial, relu) -> D_local = act(partial + adj @ table)."""
    xw: Callable
    spmm: Callable
    spmm_partial: Optional[Callable] = None
    spmm_finish: Optional[Callable] = None
    gat: Optional[Callable] = None      # gat(adj_compact, table, attention, alpha, relu) -> D_local (edge softmax)
    xw_act: Optional[Callable] = None   # xw_act(Z_local, Wt, relu) -> act(Z.W): second stage of the aggregate-first order


def hip_backend():
    from . import ops

    def xw(fea, Wt):
        if isinstance(fea, ops.Csr):
            W = ops.transpose(Wt, ldo=Wt.shape[0])                    # B [P, M] -> W [M, P]
            return ops.spmm(fea, W, relu=False)
        return ops.xw_dense(fea, Wt)

    return Backend(xw=xw, spmm=lambda adj, table, relu: ops.spmm(adj, table, relu=relu),
                   spmm_partial=lambda adj, table: ops.spmm_acc(adj, table, partial_out=True),
                   spmm_finish=lambda adj, table, partial, relu: ops.spmm_acc(adj, table, relu=relu, acc_in=partial),
                   gat=lambda adj, table, att, alpha, relu: ops.gat_aggregate(adj, table, att, alpha=alpha, relu=relu,
                                                                              fill_dead_rows=False),
                   xw_act=lambda Z, Wt, relu: ops.xw_dense(Z, Wt, relu=relu))


def _stages(backend: Backend, fea_local, Wt, relu, aggregate_first):
    """(rows to exchange, relu of the aggregation, what follows the aggregation).  Reference order: the rows of
    H = X.W travel and the aggregation applies the activation.  aggregate_first (dense X narrower than the output,
    sgx_layer_desc.order): the rows of X travel -- M_fea instead of P columns over xGMI and per gathered edge --
    and the weight product with the activation follows the aggregation."""
    if not aggregate_first:
        return backend.xw(fea_local, Wt), relu, lambda d: d
    if backend.xw_act is None or not isinstance(fea_local, torch.Tensor):
        raise ValueError("aggregate_first needs dense local features and a backend with xw_act")
    return fea_local, False, lambda z: backend.xw_act(z, Wt, relu)


def layer_allgather(backend: Backend, adj_local, fea_local, Wt, relu, bounds, group=None, h_global=None,
                    aggregate_first=False):
    """adj_local: rows of this rank, GLOBAL column indices.  Returns D_local."""
    world = len(bounds) - 1
    h_local, relu, finish = _stages(backend, fea_local, Wt, relu, aggregate_first)
    sizes = [bounds[g + 1] - bounds[g] for g in range(world)]
    P = h_local.shape[1]
    if h_global is None:
        h_global = torch.empty((bounds[-1], P), dtype=h_local.dtype, device=h_local.device)
    if len(set(sizes)) == 1:
        all_gather_into(h_global, h_local.contiguous(), group=group)
    else:
        # unequal blocks (nnz-balanced partition): gather blocks padded to the largest, then compact
        big = max(sizes)
        padded = h_local.new_zeros((big, P))
        padded[:h_local.shape[0]] = h_local
        stage = h_local.new_empty((world * big, P))
        all_gather_into(stage, padded, group=group)
        for g in range(world):
            h_global[bounds[g]:bounds[g + 1]] = stage[g * big:g * big + sizes[g]]
    return finish(backend.spmm(adj_local, h_global, relu))


def layer_halo(backend: Backend, adj_compact, fea_local, Wt, relu, plan: HaloPlan, group=None, table=None,
               attention=None, alpha=0.2, aggregate_first=False):
    """adj_compact: rows of this rank with column indices already remapped by build_halo_plan.
    attention: the GAT vector a [2P] -> the edge-softmax aggregate instead of A.H.  The softmax is
    row-local, so the same halo rows serve it: the scores Wh.a2 of the halo rows are recomputed from
    the received rows, own row r is row r of the compact table.  (Rows left without a live edge give
    0 here: the dense emulation's mean over ALL nodes would need one more reduction across ranks.)"""
    if aggregate_first and attention is not None:
        raise ValueError("the edge softmax needs Wh: no aggregate-first order for GAT")
    h_local, relu, finish = _stages(backend, fea_local, Wt, relu, aggregate_first)
    P = h_local.shape[1]
    if table is None:
        table = torch.empty((plan.n_table, P), dtype=h_local.dtype, device=h_local.device)
    table[:plan.n_own] = h_local
    packed = h_local.index_select(0, plan.send_rows) if plan.send_rows.numel() else h_local.new_empty((0, P))
    all_to_all_rows(table[plan.n_own:], packed, plan.recv_counts, plan.send_counts, group=group)
    if attention is not None:
        return backend.gat(adj_compact, table, attention, alpha, relu)
    return finish(backend.spmm(adj_compact, table, relu))


def split_own_halo(rowptr, col_compact, val, n_own):
    """Cuts a rank's CSR rows into the edges whose column is an own row (index < n_own) and the edges
    that reference halo rows (renumbered from 0); the order of edges inside a row is kept.
    Returns ((rowptr, col, val) own, (rowptr, col, val) halo)."""
    n_rows = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(n_rows, device=rowptr.device), deg)
    own = col_compact < n_own
    parts = []
    for mask, shift in ((own, 0), (~own, n_own)):
        cnt = torch.bincount(row[mask], minlength=n_rows)
        rp = torch.zeros(n_rows + 1, dtype=torch.int32, device=rowptr.device)
        rp[1:] = torch.cumsum(cnt, 0)
        parts.append((rp, (col_compact[mask] - shift).to(torch.int32).contiguous(), val[mask].contiguous()))
    return parts[0], parts[1]


def layer_halo_overlap(backend: Backend, adj_own, adj_halo, fea_local, Wt, relu, plan: HaloPlan, group=None,
                       halo_table=None, aggregate_first=False):
    """The halo exchange hidden behind the aggregation of the own-partition edges: start the
    all-to-all of the halo rows, sum the own edges into fp32 partials meanwhile, wait, add the halo
    edges (the PIPO overlap of the reference, K.cpp:3651-3749, moved to the inter-GPU step).
    adj_own: columns = own row ids; adj_halo: columns = rows of the received halo table."""
    h_local, relu, finish = _stages(backend, fea_local, Wt, relu, aggregate_first)
    P = h_local.shape[1]
    n_halo = sum(plan.recv_counts)
    if halo_table is None:
        halo_table = torch.empty((n_halo, P), dtype=h_local.dtype, device=h_local.device)
    packed = h_local.index_select(0, plan.send_rows) if plan.send_rows.numel() else h_local.new_empty((0, P))
    staged = _host_staged(packed, group) or (halo_table.is_cuda and dist.get_backend(group) == "gloo")
    work = None
    if staged or dist.get_backend(group) == "gloo":
        all_to_all_rows(halo_table[:n_halo], packed, plan.recv_counts, plan.send_counts, group=group)      # no async on gloo
    else:
        work = dist.all_to_all_single(halo_table[:n_halo], packed, output_split_sizes=plan.recv_counts,
                                      input_split_sizes=plan.send_counts, group=group, async_op=True)
    partial = backend.spmm_partial(adj_own, h_local)          # a view with padded rows is fine: the kernels take a row pitch
    if work is not None:
        work.wait()
    return finish(backend.spmm_finish(adj_halo, halo_table, partial, relu))
