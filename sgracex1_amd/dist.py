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


def all_reduce_sum(t, group=None):
    """In-place sum over the ranks (a handful of floats: the column sums behind the GAT layer's dead-row rule)."""
    if _host_staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def all_gather_direct(out, inp, bounds, rank, group=None):
    """The all-gather of H as one batch of point-to-point transfers: every rank sends its block to every peer and
    receives every peer's block straight into its place -- (world - 1) concurrent transfers per rank, one per xGMI
    link of the fully connected node ('direct' rather than ring, whose 7 serial steps each wait on one
    link).  RCCL runs the batch as one group of send/recv pairs."""
    world = len(bounds) - 1
    lo, hi = bounds[rank], bounds[rank + 1]
    staged = _host_staged(inp, group) or (out.is_cuda and dist.get_backend(group) == "gloo")
    src = inp.cpu().contiguous() if staged else inp.contiguous()
    dst = torch.empty(out.shape, dtype=out.dtype) if staged else out
    ops_ = []
    for step in range(1, world):                      # peer order staggered by rank: no two ranks start on the same target
        to, frm = (rank + step) % world, (rank - step) % world
        ops_.append(dist.P2POp(dist.isend, src, to, group=group))
        ops_.append(dist.P2POp(dist.irecv, dst[bounds[frm]:bounds[frm + 1]], frm, group=group))
    for w in (dist.batch_isend_irecv(ops_) if ops_ else []):
        w.wait()
    dst[lo:hi] = src
    if staged:
        out.copy_(dst)


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
    send_rows32: Optional[torch.Tensor] = None    # the same ids as int32 (what the pack kernel reads)

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
    send_rows = (req_in - lo).contiguous()
    return HaloPlan(bounds, rank, col_compact.to(torch.int32), send_rows, send_counts, recv_counts, hi - lo,
                    send_rows32=send_rows.to(torch.int32))


def any_rank_has_dead_rows(adj_local, group=None, device=None):
    """GAT: does ANY rank hold a row without a live edge?  That depends on the adjacency VALUES (`val > 0`, e.g. what a
    quantiser left of them), not on the structure the halo plan describes, so the answer is kept on the adjacency object
    (with the version counter of its value tensor: an in-place change asks again), never on the plan -- a plan reused
    with another adjacency would otherwise keep a stale answer and rows without a neighbour would silently get 0 instead
    of the all-node mean (SG.py:638-641).  Every rank takes its branch from the ALL-REDUCED flag only; the ranks call
    this with adjacency objects made in lockstep (as every collective here assumes), so they hit or miss together."""
    val = getattr(adj_local, "val", None)
    version = getattr(val, "_version", None)
    key = ("any_dead_rows", id(group))
    store = getattr(adj_local, "__dict__", None)
    hit = store.get("_sgx_dist_cache", {}).get(key) if store is not None else None
    if hit is not None and hit[0] == version:
        return hit[1]
    local = bool(getattr(adj_local, "has_dead_rows", True))          # an adjacency that cannot say: assume so
    dev = device if device is not None else (val.device if isinstance(val, torch.Tensor) else "cpu")
    flag = torch.tensor([float(local)], dtype=torch.float32, device=dev)
    answer = bool(all_reduce_sum(flag, group=group).item() > 0)
    if store is not None:
        store.setdefault("_sgx_dist_cache", {})[key] = (version, answer)
    return answer


@dataclass
class Backend:
    """Local compute: xw(fea_local, Wt) -> H_local [n_local, P];  spmm(adj_csr, table, relu) -> D_local;
    for the overlapped exchange also spmm_partial(adj, table) -> fp32 sums and
    spmm_finish(adj, table, partial, relu) -> D_local = act(partial + adj @ table)."""
    xw: Callable
    spmm: Callable
    spmm_partial: Optional[Callable] = None
    spmm_finish: Optional[Callable] = None
    gat: Optional[Callable] = None      # gat(adj_compact, table, attention, alpha, relu, fill_row, n_nodes) -> D_local
    xw_act: Optional[Callable] = None   # xw_act(Z_local, Wt, relu) -> act(Z.W): second stage of the aggregate-first order
    pack: Optional[Callable] = None     # pack(h_local, send_rows32) -> rows gathered into a send buffer, on the current stream
    col_sums: Optional[Callable] = None  # col_sums(h_local) -> fp32 [P] column sums of the rank's rows


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
                   gat=lambda adj, table, att, alpha, relu, fill_row=None, n_nodes=None: ops.gat_aggregate(
                       adj, table, att, alpha=alpha, relu=relu, fill_dead_rows=False, fill_row=fill_row, n_nodes=n_nodes),
                   xw_act=lambda Z, Wt, relu: ops.xw_dense(Z, Wt, relu=relu),
                   pack=lambda h, rows32: ops.pack_rows(h, rows32), col_sums=lambda h: ops.col_sums(h))


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


def _pack(backend: Backend, h_local, plan: HaloPlan):
    """The rows the peers asked for, gathered into one send buffer (HIP pack kernel; index_select without one)."""
    if not plan.send_rows.numel():
        return h_local.new_empty((0, h_local.shape[1]))
    if backend.pack is not None and plan.send_rows32 is not None:
        return backend.pack(h_local, plan.send_rows32)
    return h_local.index_select(0, plan.send_rows)


def layer_allgather(backend: Backend, adj_local, fea_local, Wt, relu, bounds, group=None, h_global=None,
                    aggregate_first=False, direct=False, rank=None):
    """adj_local: rows of this rank, GLOBAL column indices.  Returns D_local.
    direct: the gather as one batch of point-to-point transfers (all_gather_direct) instead of the library's
    all_gather_into_tensor."""
    world = len(bounds) - 1
    h_local, relu, finish = _stages(backend, fea_local, Wt, relu, aggregate_first)
    sizes = [bounds[g + 1] - bounds[g] for g in range(world)]
    P = h_local.shape[1]
    if h_global is None:
        h_global = torch.empty((bounds[-1], P), dtype=h_local.dtype, device=h_local.device)
    if direct:
        all_gather_direct(h_global, h_local, bounds, dist.get_rank(group) if rank is None else rank, group=group)
    elif len(set(sizes)) == 1:
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
               attention=None, alpha=0.2, aggregate_first=False, fill_dead_rows=None):
    """adj_compact: rows of this rank with column indices already remapped by build_halo_plan.
    attention: the GAT vector a [2P] -> the edge-softmax aggregate instead of A.H.  The softmax is
    row-local, so the same halo rows serve it: the scores Wh.a2 of the halo rows are recomputed from
    the received rows, own row r is row r of the compact table.
    Rows without a live edge (GAT): the reference's masked dense row is constant, its softmax uniform over ALL
    nodes, the row receives the mean of all rows of Wh (SG.py:638-641).  One rank sees only its own and its halo
    rows, so the column sums of every rank's rows are all-reduced (P floats) and the mean is handed to the
    aggregate -- the same result as on one GPU.  fill_dead_rows: None = do this when any rank holds such a row
    (decided once per ADJACENCY with one all-reduce of a flag, any_rank_has_dead_rows), True / False = always / never
    (then such rows give 0)."""
    if aggregate_first and attention is not None:
        raise ValueError("the edge softmax needs Wh: no aggregate-first order for GAT")
    h_local, relu, finish = _stages(backend, fea_local, Wt, relu, aggregate_first)
    P = h_local.shape[1]
    if table is None:
        table = torch.empty((plan.n_table, P), dtype=h_local.dtype, device=h_local.device)
    table[:plan.n_own] = h_local
    packed = _pack(backend, h_local, plan)
    all_to_all_rows(table[plan.n_own:], packed, plan.recv_counts, plan.send_counts, group=group)
    if attention is not None:
        if fill_dead_rows is None:
            fill_dead_rows = any_rank_has_dead_rows(adj_compact, group=group, device=h_local.device)
        if not fill_dead_rows:
            return backend.gat(adj_compact, table, attention, alpha, relu)
        if backend.col_sums is None:
            raise ValueError("rows without a live edge need a backend with col_sums (the mean row of all nodes)")
        sums = all_reduce_sum(backend.col_sums(h_local).to(torch.float32), group=group)
        n_nodes = plan.bounds[-1]
        return backend.gat(adj_compact, table, attention, alpha, relu, sums / float(n_nodes), n_nodes)
    return finish(backend.spmm(adj_compact, table, relu))


_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


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
    on_gloo = dist.get_backend(group) == "gloo"
    work = None
    if on_gloo or not h_local.is_cuda:
        # no asynchronous collective on gloo: pack, exchange, then aggregate (the rehearsal path)
        all_to_all_rows(halo_table[:n_halo], _pack(backend, h_local, plan), plan.recv_counts, plan.send_counts, group=group)
    else:
        # pack kernel and collective on a side stream: the aggregation of the own-partition edges below starts on
        # the compute stream right away instead of behind the pack
        side = _side_stream(h_local.device)
        side.wait_stream(torch.cuda.current_stream())          # H is complete
        with torch.cuda.stream(side):
            packed = _pack(backend, h_local, plan)
            work = dist.all_to_all_single(halo_table[:n_halo], packed, output_split_sizes=plan.recv_counts,
                                          input_split_sizes=plan.send_counts, group=group, async_op=True)
    partial = backend.spmm_partial(adj_own, h_local)          # a view with padded rows is fine: the kernels take a row pitch
    if work is not None:
        work.wait()
    return finish(backend.spmm_finish(adj_halo, halo_table, partial, relu))
