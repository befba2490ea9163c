"""Torch-facing wrappers over the C ABI (include/sgx.h).

PyTorch is plumbing here: it owns device memory and the stream.  Every function passes raw
device pointers to libsgx.so and returns torch tensors that alias buffers the call filled.
All tensors must live on a ROCm device ("cuda"); nothing in this module computes on the CPU.
"""
import ctypes

import torch

from . import _lib
from ._lib import (SGX_ACC_F32, SGX_ACC_REF_HALF, SGX_F16, SGX_F32, SGX_ORDER_AGGREGATE_FIRST, SGX_ORDER_REFERENCE,
                   LayerDesc, check, lib)

_DTYPES = {torch.float16: SGX_F16, torch.float32: SGX_F32}


def dtype_code(dtype):
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise TypeError(f"sgx supports float16 and float32 element types, got {dtype}") from None


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a tensor on the GPU (got {type(t).__name__}"
                         f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _dev2d(t, name):
    """2-D tensor whose rows may be padded (row stride >= width, unit column stride)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a tensor on the GPU")
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name} must be 2-D with unit column stride")
    return t


def _out(out, rows, cols, dtype, device, name="out"):
    """A caller-provided result buffer: [rows, >= cols] of the right element type on the right device, unit
    column stride (rows may be padded).  The library writes through the raw pointer, so a wrong buffer must
    be refused here."""
    if out is None:
        return torch.empty((rows, cols), dtype=dtype, device=device)
    _dev2d(out, name)
    if out.dtype != dtype or out.device != device or out.shape[0] != rows or out.shape[1] != cols:
        raise ValueError(f"{name} must be a [{rows}, {cols}] {dtype} tensor on {device} "
                         f"(got {tuple(out.shape)} {out.dtype} on {out.device})")
    return out


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def cached_on(tensor, key, build):
    """Derived data (a CSR, a transposed pattern ...) kept ON the tensor object it was built from, so that it
    lives exactly as long as that tensor and can never be handed to another tensor that happens to reuse the
    address; an in-place change of the tensor (its version counter) rebuilds it."""
    store = tensor.__dict__.setdefault("_sgx_cache", {})
    hit = store.get(key)
    if hit is None or hit[0] != tensor._version:
        hit = store[key] = (tensor._version, build())
    return hit[1]


_workspaces = {}


def _workspace(device, nbytes):
    """Grow-only scratch per (device, stream); the library itself never allocates."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


class Plan:
    """Row schedule of one CSR matrix (sgx_plan): which rows are split across wavefronts."""

    def __init__(self, rowptr, long_threshold=0, chunk=0):
        _dev(rowptr, "rowptr")
        h = ctypes.c_void_p()
        check(lib.sgx_plan_create_ex(ctypes.byref(h), _ptr(rowptr), rowptr.numel() - 1, int(long_threshold), int(chunk),
                                     _stream()), "sgx_plan_create_ex")
        self._h = h
        self.n_rows = rowptr.numel() - 1

    @property
    def handle(self):
        return self._h

    @property
    def long_rows(self):
        return lib.sgx_plan_long_rows(self._h)

    @property
    def long_threshold(self):
        return lib.sgx_plan_long_threshold(self._h)

    @property
    def natural_utilization(self):
        return lib.sgx_plan_natural_utilization(self._h)

    @property
    def reordered(self):
        return bool(lib.sgx_plan_reordered(self._h))

    ARRAYS = ("long_row", "long_first", "task_row", "task_e0", "task_e1", "row_order", "win_order", "scan_win")

    def export(self, name):
        """One of the schedule's device arrays (ARRAYS) as an int32 tensor -- for inspection and tests."""
        which = self.ARRAYS.index(name)
        n = lib.sgx_plan_export(self._h, which, None, 0, _stream())
        if n < 0:
            check(int(n), "sgx_plan_export")
        out = torch.empty(int(n), dtype=torch.int32, device="cuda")
        if n:
            got = lib.sgx_plan_export(self._h, which, _ptr(out), int(n), _stream())
            if got < 0:
                check(int(got), "sgx_plan_export")
        if name == "win_order":                 # bytes: the row (0..63) of every rank of every 64-row window
            return out.view(torch.uint8)
        return out

    def __del__(self, _destroy=lib.sgx_plan_destroy):        # bound at definition: module globals are gone at shutdown
        h, self._h = getattr(self, "_h", None), None
        if h:
            _destroy(h)


class Csr:
    """CSR matrix in HBM: rowptr[int32, n_rows+1], col[int32, nnz], val[f16|f32, nnz]."""

    def __init__(self, rowptr, col, val, n_cols, plan=None):
        self.rowptr = _dev(rowptr, "rowptr")
        self.col = _dev(col, "col")
        self.val = _dev(val, "val")
        if rowptr.dtype != torch.int32 or col.dtype != torch.int32:
            raise TypeError("CSR indices must be int32 (the reference's `int` ports, K.cpp:3769-3773)")
        self.n_rows = rowptr.numel() - 1
        self.n_cols = int(n_cols)
        self._nnz = col.numel()
        if self._nnz == 0:
            # an empty tensor has no address; the library wants non-NULL index / value arrays even when
            # rowPtr says there is nothing to read
            self.col = torch.zeros(1, dtype=torch.int32, device=self.rowptr.device)
            self.val = torch.zeros(1, dtype=val.dtype, device=self.rowptr.device)
        self._plan = plan
        self._dead_rows = None
        self._quantized = {}

    @property
    def has_dead_rows(self):
        """True when some row holds no positive value -- the rows the GAT mask `adj > 0` (SG.py:640)
        leaves without a neighbour.  One device->host sync, once per matrix."""
        if self._dead_rows is None:
            deg = (self.rowptr[1:] - self.rowptr[:-1]).long()
            row = torch.repeat_interleave(torch.arange(self.n_rows, device=self.val.device), deg)
            live = torch.zeros(self.n_rows, dtype=torch.int32, device=self.val.device)
            live.index_add_(0, row, (self.val[:self.nnz] > 0).to(torch.int32))
            self._dead_rows = bool((live == 0).any().item())
        return self._dead_rows

    def quantized(self, qc):
        """The adjacency on the unsigned w_qbits grid (SG.py:626), quantised once per graph and constants."""
        key = (qc.w_qbits, qc.a_s, qc.a_z)
        if key not in self._quantized:
            self._quantized[key] = Csr(self.rowptr, self.col, fake_quantize(self.val, 0, qc.w_qbits, qc.a_s, qc.a_z),
                                       self.n_cols, self._plan)
        return self._quantized[key]

    @property
    def nnz(self):
        return self._nnz

    @property
    def plan(self):
        if self._plan is None:
            self._plan = Plan(self.rowptr)
        return self._plan

    @property
    def gat_plan(self):
        """The schedule for the edge-softmax aggregate: hub rows cut at 256 edges.  Its first stage (the softmax weights,
        csrc/gat.hip, gat_scan.hip) keeps a row of up to 256 edges in registers -- per row, or per window of stored entries
        on a plan in degree order (the plan's scan_win, built for this cut) -- longer rows go through the plan's tasks.  Measured on
        R-MAT graphs of 2.4 M / 7.5 M / 29 M edges (tools/plan_cut_probe.py): 256 is the best or within 2 % of it for one
        head and for 8, while the plain aggregation prefers 512 / 1024 / 2048 (Plan's default).  The plan also tells the
        library the stored-entry count it sizes the weights with."""
        if getattr(self, "_gat_plan", None) is None:
            self._gat_plan = Plan(self.rowptr, 256, 256)
        return self._gat_plan

    @property
    def wants_plan(self):
        """Building a plan costs one device->host copy and a stream sync; matrices this small finish
        in microseconds on any schedule, so they run without one unless a plan already exists."""
        return self._plan is not None or self.nnz >= 8192

    def to(self, dtype):
        # (the schedule depends on rowptr only: the copy shares this matrix's -- built here if it is wanted and not there
        # yet, so that a copy made per training step does not build one per step)
        return self if self.val.dtype == dtype else Csr(self.rowptr, self.col, self.val.to(dtype), self.n_cols,
                                                        self.plan if self.wants_plan else None)

    def validate(self):
        check(lib.sgx_csr_validate(_ptr(self.rowptr), _ptr(self.col), self.n_rows, self.n_cols, self.nnz, _stream()),
              "sgx_csr_validate")

    @staticmethod
    def from_dense(dense, dtype=None):
        """Same CSR torch's `_to_sparse_csr()` yields in the molecule notebook (MOL cell 18)."""
        sp = dense.to_sparse_csr()
        val = sp.values() if dtype is None else sp.values().to(dtype)
        return Csr(sp.crow_indices().to(torch.int32).contiguous(), sp.col_indices().to(torch.int32).contiguous(),
                   val.contiguous(), dense.shape[1])

    @staticmethod
    def from_coo(row, col, val, n_rows, n_cols):
        """Edges sorted by row (SG.py ships COO: rowPtr_adj_buffer holds row indices, SG.py:1245)."""
        _dev(row, "row")
        rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=row.device)
        check(lib.sgx_coo_to_csr(_ptr(row.to(torch.int32).contiguous()), row.numel(), n_rows, _ptr(rowptr), _stream()),
              "sgx_coo_to_csr")
        return Csr(rowptr, col.to(torch.int32).contiguous(), val.contiguous(), n_cols)


def csr_from_edge_index(edge_index, n_rows, n_cols=None, values=None, dtype=torch.float16):
    """[2, E] edge list (any order, duplicates add up) -> Csr, without the dense N x N detour of
    `to_dense_adj(...)._to_sparse_csr()` (MOL cell 18): sort by (row, col), merge duplicates,
    row pointer by sgx_coo_to_csr.  Gives the same CSR as the dense route."""
    _dev(edge_index, "edge_index")
    n_cols = n_rows if n_cols is None else n_cols
    key = edge_index[0].to(torch.int64) * n_cols + edge_index[1].to(torch.int64)
    w = torch.ones(key.numel(), dtype=torch.float32, device=key.device) if values is None else values.float()
    ukey, inv = torch.unique(key, return_inverse=True)
    val = torch.zeros(ukey.numel(), dtype=torch.float32, device=key.device).index_add_(0, inv, w)
    row = torch.div(ukey, n_cols, rounding_mode="floor")
    col = ukey - row * n_cols
    keep = val != 0                                     # a dense matrix cannot hold explicit zeros either
    return Csr.from_coo(row[keep].to(torch.int32), col[keep].to(torch.int32), val[keep].to(dtype), n_rows, n_cols)


def _gatherable(H, n_feat=None, nnz=0):
    """The table as the aggregation wants it: rows that start on a dword take 16-byte gathers; a table whose
    rows start on odd halves (47 fp16 columns, unpadded) would be gathered one element per lane, so it is
    copied once into rows of table_pitch() elements -- N x P elements moved against E x P gathered.  A large
    table whose pitch lets rows straddle 128-byte lines (100 halves) is copied too when every row is gathered
    often enough (32 edges per table row) for the 18-25 % the gathers gain to outweigh the copy."""
    n_feat = H.shape[1] if n_feat is None else n_feat
    if (H.stride(0) * H.element_size()) % 4 == 0 and H.data_ptr() % 4 == 0:
        straddles = H.stride(0) != table_pitch(n_feat, H.element_size()) and (H.stride(0) * H.element_size()) % 128 != 0
        if not (straddles and H.shape[0] * n_feat >= (1 << 22) and nnz >= 32 * H.shape[0]):
            return H
    elif H.shape[0] * n_feat < (1 << 16):
        return H
    padded = torch.empty((H.shape[0], table_pitch(n_feat, H.element_size())), dtype=H.dtype, device=H.device)
    padded[:, :n_feat] = H[:, :n_feat]
    return padded


def spmm(adj, H, relu=False, n_feat=None, out=None, use_plan=True, acc_mode=SGX_ACC_F32, spmm_block=1):
    """D = act(A @ H[:, :n_feat]) -- the aggregation stage alone (sgx_spmm_csr)."""
    _dev2d(H, "H")
    n_feat = H.shape[1] if n_feat is None else n_feat
    if acc_mode == SGX_ACC_F32:
        H = _gatherable(H, n_feat, adj.nnz)
    code = dtype_code(H.dtype)
    if adj.val.dtype != H.dtype:
        raise TypeError("adjacency values and H must share one element type (MM.h:129-139)")
    if H.shape[0] < adj.n_cols:
        raise ValueError(f"H has {H.shape[0]} rows, the adjacency refers to {adj.n_cols} columns")
    out = _out(out, adj.n_rows, n_feat, H.dtype, H.device)
    plan = adj.plan if (use_plan and adj.wants_plan) else None
    sbytes = lib.sgx_spmm_scratch_bytes(plan.handle, n_feat) if plan is not None else 0
    scratch = _workspace(H.device, sbytes) if sbytes else None
    check(lib.sgx_spmm_csr(code, acc_mode, spmm_block, int(bool(relu)), adj.n_rows, H.shape[0], n_feat,
                           _ptr(adj.rowptr), _ptr(adj.col), _ptr(adj.val), _ptr(H), H.stride(0),
                           _ptr(out), out.stride(0), plan.handle if plan is not None else None,
                           _ptr(scratch), sbytes, _stream()), "sgx_spmm_csr")
    return out


def xw_sparse(X, W, out=None, use_plan=True):
    """H = X @ W for a CSR X and a row-major W [M_fea, P] -- the X.W stage alone in gemm_mode 0 (sgx_xw_sparse):
    the weight slice resident in LDS for a large X, gathered through L2 otherwise; same sums either way."""
    _dev2d(W, "W")
    if X.val.dtype != W.dtype:
        raise TypeError("feature values and W must share one element type (MM.h:129-139)")
    if W.shape[0] < X.n_cols:
        raise ValueError(f"W has {W.shape[0]} rows, X refers to {X.n_cols} columns")
    P = W.shape[1]
    W = _gatherable(W, P, X.nnz)
    out = _out(out, X.n_rows, P, W.dtype, W.device)
    plan = X.plan if (use_plan and X.wants_plan) else None
    sbytes = lib.sgx_spmm_scratch_bytes(plan.handle, P) if plan is not None else 0
    scratch = _workspace(W.device, sbytes) if sbytes else None
    check(lib.sgx_xw_sparse(dtype_code(W.dtype), SGX_ACC_F32, 1, X.n_rows, X.n_cols, P, _ptr(X.rowptr), _ptr(X.col),
                            _ptr(X.val), _ptr(W), W.stride(0), _ptr(out), out.stride(0),
                            plan.handle if plan is not None else None, _ptr(scratch), sbytes, _stream()), "sgx_xw_sparse")
    return out


def spmm_acc(adj, H, relu=False, acc_in=None, partial_out=False, out=None, use_plan=True):
    """Two-pass aggregation (sgx_spmm_csr_acc): with partial_out the fp32 sums acc_in + A @ H are
    returned; otherwise D = act(acc_in + A @ H) in H's dtype."""
    _dev2d(H, "H")
    n_feat = H.shape[1]
    H = _gatherable(H, n_feat, adj.nnz)
    code = dtype_code(H.dtype)
    if adj.val.dtype != H.dtype:
        raise TypeError("adjacency values and H must share one element type (MM.h:129-139)")
    if H.shape[0] < adj.n_cols:
        raise ValueError(f"H has {H.shape[0]} rows, the adjacency refers to {adj.n_cols} columns")
    if acc_in is not None:
        _dev(acc_in, "acc_in")
        if acc_in.dtype != torch.float32 or acc_in.shape != (adj.n_rows, n_feat):
            raise ValueError("acc_in must be float32 [n_rows, n_feat]")
    acc_out = torch.empty((adj.n_rows, n_feat), dtype=torch.float32, device=H.device) if partial_out else None
    if not partial_out:
        out = _out(out, adj.n_rows, n_feat, H.dtype, H.device)
    plan = adj.plan if (use_plan and adj.wants_plan) else None
    sbytes = lib.sgx_spmm_scratch_bytes(plan.handle, n_feat) if plan is not None else 0
    scratch = _workspace(H.device, sbytes) if sbytes else None
    check(lib.sgx_spmm_csr_acc(code, int(bool(relu)), adj.n_rows, H.shape[0], n_feat, _ptr(adj.rowptr), _ptr(adj.col),
                               _ptr(adj.val), _ptr(H), H.stride(0), None if partial_out else _ptr(out),
                               0 if partial_out else out.stride(0), _ptr(acc_in), _ptr(acc_out), n_feat,
                               plan.handle if plan is not None else None, _ptr(scratch), sbytes, _stream()),
          "sgx_spmm_csr_acc")
    return acc_out if partial_out else out


def table_pitch(width, elem_size):
    """Row pitch (elements) for a table the aggregation gathers from: 16-byte multiples, whole 128-byte lines
    where that costs at most a third more bytes, powers of two below one line (sgx_ldh, csrc/sgx_internal.h)."""
    row = (width * elem_size + 15) // 16 * 16
    if row < 128:
        pitch = 16
        while pitch < row:
            pitch *= 2
    else:
        lines = (row + 127) // 128 * 128
        pitch = lines if 3 * lines <= 4 * row else row
    return pitch // elem_size


def xw_dense(X, Wt, ldh=None, acc_mode=SGX_ACC_F32, spmm_block=1, relu=False):
    """H = X @ Wt.T on the matrix cores (sgx_xw_dense).  Wt = weights transposed, [P, M].
    relu: the activation on the stores (sgx_xw_dense_act; the second stage of the aggregate-first order)."""
    _dev2d(X, "X")
    _dev2d(Wt, "Wt")
    code = dtype_code(X.dtype)
    if Wt.dtype != X.dtype or X.shape[1] != Wt.shape[1]:
        raise TypeError(f"X [n, M] and Wt [P, M] must share the element type and M (got {X.dtype} {tuple(X.shape)}, "
                        f"{Wt.dtype} {tuple(Wt.shape)})")
    P, M = Wt.shape
    ldh = table_pitch(P, X.element_size()) if ldh is None else ldh
    H = torch.empty((X.shape[0], ldh), dtype=X.dtype, device=X.device)
    if relu:
        if acc_mode != SGX_ACC_F32:
            raise ValueError("the activation rides on the fp32-accumulate kernels only")
        check(lib.sgx_xw_dense_act(code, 1, X.shape[0], M, P, _ptr(X), X.stride(0), _ptr(Wt), Wt.stride(0), _ptr(H), ldh,
                                   _stream()), "sgx_xw_dense_act")
        return H[:, :P]
    check(lib.sgx_xw_dense(code, acc_mode, spmm_block, X.shape[0], M, P, _ptr(X), X.stride(0), _ptr(Wt), Wt.stride(0),
                           _ptr(H), ldh, _stream()), "sgx_xw_dense")
    return H[:, :P]


def transpose(x, ldo=None):
    _dev(x, "x")
    rows, cols = x.shape
    ldo = rows if ldo is None else ldo
    out = torch.empty((cols, ldo), dtype=x.dtype, device=x.device)
    check(lib.sgx_transpose(dtype_code(x.dtype), rows, cols, _ptr(x), x.stride(0), _ptr(out), ldo, _stream()),
          "sgx_transpose")
    return out


def layer_forward(adj, fea, Wt, relu=False, gat_attention=None, alpha=0.2, want_edge_outputs=False, quant_int8=False,
                  acc_mode=SGX_ACC_F32, spmm_block=1, bias_count=0, out=None, use_plan=True, agg_events=None,
                  quant=None, adj_quantized=False, cache_quantized_adj=True, fea_threads=1, adj_threads=1,
                  gat_heads=1, order="reference"):
    """One fused layer  D = act(A . (X . W))  through sgx_layer_forward.

    adj : Csr [N, M_adj];  fea : Csr [M_adj, M_fea] (gemm_mode 0) or dense tensor (gemm_mode 1);
    Wt  : [P, M_fea] -- the weights TRANSPOSED, what the reference writes into B_buffer.
    Returns D [N, P] (and (E, S) per-edge tensors when want_edge_outputs with GAT).
    quant: a quant.QuantConstants -- run the layer with the quantised arithmetic of the SGRACE
    bitstream (fp32 tensors only); quant_int8: with dense features, X and W go to the int8 matrix cores as the integer
    codes of their grids (SGX_QUANT_INT8: exact int32 sums, X read as bytes), "auto" = where that is the faster form
    (SGX_QUANT_INT8_AUTO: M_fea > 128); adj_quantized: adj.val already went through the quantiser;
    cache_quantized_adj: quantise the adjacency once per graph on the host side instead of inside
    every call (always done for GAT, whose mask decides how rows without a live edge are treated).
    order: "reference" -- X.W first, as the reference's dataflow; "aggregate_first" -- D = act((A.X).W), which
    gathers M_fea instead of P columns per edge (dense X, GCN aggregate, default arithmetic only); "auto" --
    aggregate first where that is allowed and M_fea < P.
    """
    _dev(Wt, "Wt")
    code = dtype_code(Wt.dtype)
    P, M_fea = Wt.shape
    d = LayerDesc()
    gemm_mode = 0 if isinstance(fea, Csr) else 1
    if order not in ("reference", "aggregate_first", "auto"):
        raise ValueError(f"order must be 'reference', 'aggregate_first' or 'auto', not {order!r}")
    can_swap = gemm_mode == 1 and gat_attention is None and quant is None and acc_mode == SGX_ACC_F32
    if order == "aggregate_first" and not can_swap:
        raise ValueError("aggregate_first needs dense features, the GCN aggregate and the default arithmetic "
                         "(no GAT, no quantised layer, no SGX_ACC_REF_HALF)")
    swap = order == "aggregate_first" or (order == "auto" and can_swap and M_fea < P)
    d.order = SGX_ORDER_AGGREGATE_FIRST if swap else SGX_ORDER_REFERENCE
    d.gemm_mode, d.relu, d.gat_mode = gemm_mode, int(bool(relu)), int(gat_attention is not None)
    d.N_adj, d.M_adj, d.M_fea, d.P_w = adj.n_rows, adj.n_cols, M_fea, P
    d.bias_count, d.dtype, d.acc_mode, d.spmm_block = bias_count, code, acc_mode, spmm_block
    d.fea_threads, d.adj_threads = int(fea_threads), int(adj_threads)      # observable in SGX_ACC_REF_HALF only
    if adj.val.dtype != Wt.dtype:
        raise TypeError("adjacency, features and weights must share one element type (MM.h:129-139)")
    if quant is not None and Wt.dtype != torch.float32:
        raise TypeError("the quantised layer works on float32 buffers (SG.py:1545)")
    if gemm_mode == 0:
        if fea.val.dtype != Wt.dtype or fea.n_rows != adj.n_cols or fea.n_cols != M_fea:
            raise ValueError("feature CSR does not match adjacency / weights")
        d.rowPtr_fea, d.columnIndex_fea, d.values_fea = (fea.rowptr.data_ptr(), fea.col.data_ptr(),
                                                         fea.val.data_ptr())
        if use_plan and fea.wants_plan:
            d.plan_fea = fea.plan.handle
    else:
        _dev(fea, "fea")
        if fea.dtype != Wt.dtype or fea.shape != (adj.n_cols, M_fea):
            raise ValueError(f"dense features must be [{adj.n_cols}, {M_fea}] {Wt.dtype}")
        d.values_fea = fea.data_ptr()
    if quant is not None and not adj_quantized and (cache_quantized_adj or gat_attention is not None):
        adj, adj_quantized = adj.quantized(quant), True
    if gat_attention is not None:
        d.gat_fill_dead_rows = int(adj.has_dead_rows)
    d.rowPtr_adj, d.columnIndex_adj, d.values_adj = adj.rowptr.data_ptr(), adj.col.data_ptr(), adj.val.data_ptr()
    if use_plan and adj.wants_plan:
        d.plan_adj = (adj.gat_plan if gat_attention is not None else adj.plan).handle
    d.B = Wt.data_ptr()
    out = _out(out, adj.n_rows, P, Wt.dtype, Wt.device)
    if out.stride(0) != P:
        raise ValueError("the layer writes D densely ([N_adj][P_w], K.cpp:802): `out` must not have padded rows")
    d.D = out.data_ptr()
    E = S = None
    if gat_attention is not None:
        att = _dev(gat_attention, "attention").reshape(-1)
        if att.numel() != 2 * P or att.dtype != Wt.dtype:
            raise ValueError("attention must hold 2*P_w elements of the layer dtype")
        d.attention, d.alpha = att.data_ptr(), float(alpha)
        if P % int(gat_heads):
            raise ValueError("P_w must be a multiple of gat_heads")
        d.gat_heads = int(gat_heads)
        if want_edge_outputs:
            es_shape = (adj.nnz,) if gat_heads == 1 else (adj.nnz, int(gat_heads))
            E = torch.empty(es_shape, dtype=torch.float32, device=Wt.device)
            S = torch.empty(es_shape, dtype=torch.float32, device=Wt.device)
            d.E, d.S = E.data_ptr(), S.data_ptr()
    if agg_events is not None:          # (begin, end) hipEvent_t handles, see hipevents.py
        d.ev_agg_begin, d.ev_agg_end = agg_events
    if quant is not None:
        qs = quant.as_struct(nnz_adj=adj.nnz, nnz_fea=fea.nnz if gemm_mode == 0 else 0, adj_done=adj_quantized)
        if quant_int8 == "auto":
            qs.flags |= _lib.SGX_QUANT_INT8_AUTO
        elif quant_int8:
            qs.flags |= _lib.SGX_QUANT_INT8
        d.quant = ctypes.pointer(qs)
    nbytes = lib.sgx_layer_workspace_bytes(ctypes.byref(d))
    ws = _workspace(Wt.device, nbytes)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib.sgx_layer_forward(ctypes.byref(d), _stream()), "sgx_layer_forward")
    return (out, E, S) if want_edge_outputs else out


def fake_quantize(x, signed, qbits, scale, zero, out=None):
    """quantization_fbits (signed) / quantization_ufbits (SG.py:238-265) of an fp32 tensor on the device."""
    _dev(x, "x")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise TypeError("fake_quantize works on contiguous float32 tensors (SG.py:1545)")
    if out is None:
        out = torch.empty_like(x)
    check(lib.sgx_fake_quantize(int(bool(signed)), int(qbits), float(1 / scale), float(zero), x.numel(), _ptr(x), _ptr(out),
                                _stream()), "sgx_fake_quantize")
    return out


def quantize_codes_i8(x, signed, qbits, scale, zero):
    """The integer codes of an fp32 matrix on its w_qbits grid as int8 [rows, pitch] (pitch = columns rounded up to
    16, pad codes 0); unsigned 8-bit codes are stored minus 128 (sgx_quantize_codes_i8).  Returns (codes, bias)."""
    _dev2d(x, "x")
    if x.dtype != torch.float32:
        raise TypeError("codes are taken from float32 values (SG.py:1545)")
    n, m = x.shape
    ldc = (m + 15) // 16 * 16
    codes = torch.empty((n, ldc), dtype=torch.int8, device=x.device)
    check(lib.sgx_quantize_codes_i8(int(bool(signed)), int(qbits), float(1 / scale), float(zero), n, m, _ptr(x), x.stride(0),
                                    _ptr(codes), ldc, _stream()), "sgx_quantize_codes_i8")
    return codes, lib.sgx_code_bias(int(bool(signed)), int(qbits))


def xw_dense_i8(Xc, Wc, M_fea, qbits, scale_fea=0, internal_bits=0):
    """H = requant((Xc . Wc^T + bias terms) / 2^(2(qbits-1))) on the int8 matrix cores (sgx_xw_dense_i8): Xc unsigned
    feature codes [n, pitch], Wc signed weight codes [P, pitch] as quantize_codes_i8 returns them."""
    n, P = Xc.shape[0], Wc.shape[0]
    H = torch.empty((n, P), dtype=torch.float32, device=Xc.device)
    ws = torch.empty(max(1, lib.sgx_xw_dense_i8_workspace_bytes(P)), dtype=torch.uint8, device=Xc.device)
    check(lib.sgx_xw_dense_i8(int(qbits), n, int(M_fea), P, _ptr(Xc), Xc.stride(0), _ptr(Wc), Wc.stride(0), int(scale_fea),
                              int(internal_bits), _ptr(H), H.stride(0), _ptr(ws), _stream()), "sgx_xw_dense_i8")
    return H


def requantize_(H, scale_fea, internal_bits):
    """H <- round_decimals(clip(H / 2^scale_fea), internal_bits - 1) in place (SG.py:607-616)."""
    _dev2d(H, "H")
    if H.dtype != torch.float32:
        raise TypeError("requantize_ works on float32")
    check(lib.sgx_requantize(H.shape[0], H.shape[1], H.stride(0), _ptr(H), int(scale_fea), int(internal_bits), _stream()),
          "sgx_requantize")
    return H


def gat_aggregate(adj, Wh, attention, alpha=0.2, relu=False, want_edge_outputs=False, fill_dead_rows=None, out=None,
                  heads=1, use_plan=True, fill_row=None, n_nodes=None):
    """Edge-softmax aggregate over an already computed Wh [adj.n_cols, F]; row r of adj is node r of Wh.
    heads > 1: F/heads columns per head, attention = heads vectors of 2*F/heads (E, S become [nnz, heads]).
    fill_dead_rows: None = decide from the adjacency (rows without a positive entry get the mean of
    all rows of Wh, as in the reference's dense emulation), False = such rows give 0.
    fill_row (fp32 [F]) with n_nodes: one rank of a partitioned graph -- dead rows receive this row (the mean over
    ALL nodes, reduced across ranks by the caller) and S = 1/n_nodes (sgx_gat_aggregate_fill)."""
    _dev2d(Wh, "Wh")
    code = dtype_code(Wh.dtype)
    N, F = Wh.shape
    if N != adj.n_cols or adj.n_rows > N:
        raise ValueError(f"Wh must have adj.n_cols = {adj.n_cols} rows (got {N}) and adj.n_rows <= adj.n_cols")
    att = _dev(attention, "attention").reshape(-1)
    if adj.val.dtype != Wh.dtype or att.dtype != Wh.dtype:
        raise TypeError("adjacency values, Wh and the attention vector must share one element type (MM.h:129-139)")
    out = _out(out, adj.n_rows, F, Wh.dtype, Wh.device)
    E = S = None
    heads = int(heads)
    if F % heads or att.numel() != 2 * F:
        raise ValueError("F must be a multiple of heads and attention must hold 2*F elements")
    if want_edge_outputs:
        es_shape = (adj.nnz,) if heads == 1 else (adj.nnz, heads)
        E = torch.empty(es_shape, dtype=torch.float32, device=Wh.device)
        S = torch.empty(es_shape, dtype=torch.float32, device=Wh.device)
    plan = adj.gat_plan.handle if (use_plan and adj.wants_plan) else None
    if fill_row is not None:
        _dev(fill_row, "fill_row")
        if fill_row.dtype != torch.float32 or fill_row.numel() != F or not n_nodes:
            raise ValueError("fill_row must be float32 [F] and come with n_nodes")
        s = torch.empty(lib.sgx_gat_scratch_bytes(N, F, heads, 0, plan) // 4, dtype=torch.float32, device=Wh.device)
        check(lib.sgx_gat_aggregate_fill(code, int(bool(relu)), adj.n_rows, N, F, heads, float(alpha), _ptr(adj.rowptr),
                                         _ptr(adj.col), _ptr(adj.val), _ptr(Wh), Wh.stride(0), _ptr(att), _ptr(out),
                                         out.stride(0), _ptr(E), _ptr(S), _ptr(fill_row.contiguous()), int(n_nodes), plan,
                                         _ptr(s), _stream()), "sgx_gat_aggregate_fill")
        return (out, E, S) if want_edge_outputs else out
    fill = int(adj.has_dead_rows if fill_dead_rows is None else bool(fill_dead_rows))
    s = torch.empty(lib.sgx_gat_scratch_bytes(N, F, heads, fill, plan) // 4, dtype=torch.float32, device=Wh.device)
    check(lib.sgx_gat_aggregate(code, int(bool(relu)), fill, adj.n_rows, N, F, heads, float(alpha), _ptr(adj.rowptr), _ptr(adj.col),
                                _ptr(adj.val), _ptr(Wh), Wh.stride(0), _ptr(att), _ptr(out), out.stride(0),
                                _ptr(E), _ptr(S), plan, _ptr(s), _stream()), "sgx_gat_aggregate")
    return (out, E, S) if want_edge_outputs else out


def col_sums(X, n_feat=None):
    """fp32 column sums of the rows of X [n, F] in a fixed order (sgx_col_sums): a rank's share of the mean row the
    partitioned GAT layer gives rows without a live edge."""
    _dev2d(X, "X")
    F = X.shape[1] if n_feat is None else n_feat
    out = torch.empty(F, dtype=torch.float32, device=X.device)
    scratch = torch.empty(lib.sgx_col_sums_scratch_bytes(F) // 4, dtype=torch.float32, device=X.device)
    check(lib.sgx_col_sums(dtype_code(X.dtype), X.shape[0], F, _ptr(X), X.stride(0), _ptr(out), _ptr(scratch), _stream()),
          "sgx_col_sums")
    return out


def pack_rows(src, row_index, out=None, stream=None):
    """out[i] = src[row_index[i]] (sgx_pack_rows): the rows of H a peer asked for, gathered into the send buffer of
    the halo exchange.  row_index int32 on the device; stream: a torch stream other than the current one to launch on."""
    _dev2d(src, "src")
    _dev(row_index, "row_index")
    if row_index.dtype != torch.int32:
        raise TypeError("row_index must be int32")
    n, F = row_index.numel(), src.shape[1]
    out = _out(out, n, F, src.dtype, src.device)
    st = ctypes.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    check(lib.sgx_pack_rows(dtype_code(src.dtype), n, F, _ptr(src), src.stride(0), _ptr(row_index), _ptr(out), out.stride(0), st),
          "sgx_pack_rows")
    return out


def xt_g(X, G):
    """grad_W = X^T @ G (sgx_xt_g): X [n, M] fp16|fp32 dense, G [n, P] fp32 -> [M, P] fp32."""
    _dev2d(X, "X")
    _dev2d(G, "G")
    if G.dtype != torch.float32:
        raise TypeError("G must be float32 (the backward pass runs in fp32, MOL cell 16)")
    n, M = X.shape
    P = G.shape[1]
    out = torch.empty((M, P), dtype=torch.float32, device=X.device)
    nbytes = lib.sgx_xt_g_workspace_bytes(n, M, P)
    ws = _workspace(X.device, nbytes)
    check(lib.sgx_xt_g(dtype_code(X.dtype), n, M, P, _ptr(X), X.stride(0), _ptr(G), G.stride(0), _ptr(out), P,
                       _ptr(ws), ws.numel(), _stream()), "sgx_xt_g")
    return out


def csr_transpose(A, return_order=False):
    """CSR of A^T (values kept, same dtype); features are fixed across epochs, so callers cache it.
    return_order: also the edge permutation (edge k of A^T is edge order[k] of A)."""
    row = torch.repeat_interleave(torch.arange(A.n_rows, device=A.col.device, dtype=torch.int64),
                                  (A.rowptr[1:] - A.rowptr[:-1]).long())
    col, val = A.col[:A.nnz], A.val[:A.nnz]
    key = col.to(torch.int64) * A.n_rows + row
    order = torch.argsort(key)
    T = Csr.from_coo(col[order].contiguous(), row[order].to(torch.int32).contiguous(), val[order].contiguous(),
                     A.n_cols, A.n_rows)
    if T.n_rows > 0 and T.nnz >= 64 * T.n_rows:
        # a feature matrix transposed: a handful of rows, each as long as the graph is wide (MUTAG: 7 rows of ~500 entries) --
        # below the size at which a plan is built by itself, and exactly the shape that needs one: without it 7 lane groups
        # walk 60 dependent steps each (58 us for a 3.4 K-entry matrix) where the plan's 64-edge tasks take a few
        T.plan
    return (T, order) if return_order else T


def gat_backward_edges(adj, E, S, G, Wh, alpha=0.2):
    """Edge pass of FPYNQ_GAT.backward (sgx_gat_backward_edges): returns (sg [nnz], g1 [n_rows]) fp32."""
    _dev2d(G, "G")
    _dev2d(Wh, "Wh")
    if G.dtype != torch.float32 or Wh.dtype != torch.float32 or E.dtype != torch.float32 or S.dtype != torch.float32:
        raise TypeError("gat_backward_edges works on float32 E, S, G, Wh (the reference's backward is fp32)")
    if Wh.shape[0] != adj.n_cols or G.shape != (adj.n_rows, Wh.shape[1]):
        raise ValueError("G must be [adj.n_rows, F] and Wh [adj.n_cols, F]")
    if Wh.stride(0) % 4 or Wh.data_ptr() % 16:           # rows are gathered 16 bytes at a time: pad them
        padded = torch.zeros((Wh.shape[0], (Wh.shape[1] + 3) // 4 * 4), dtype=torch.float32, device=Wh.device)
        padded[:, :Wh.shape[1]] = Wh
        Wh = padded[:, :Wh.shape[1]]
    sg = torch.empty(adj.nnz, dtype=torch.float32, device=G.device)
    g1 = torch.empty(adj.n_rows, dtype=torch.float32, device=G.device)
    check(lib.sgx_gat_backward_edges(dtype_code(adj.val.dtype), adj.n_rows, adj.n_cols, Wh.shape[1], float(alpha),
                                     _ptr(adj.rowptr), _ptr(adj.col), _ptr(adj.val), _ptr(E.contiguous()), _ptr(S.contiguous()),
                                     _ptr(G), G.stride(0), _ptr(Wh), Wh.stride(0), _ptr(sg), _ptr(g1), _stream()),
          "sgx_gat_backward_edges")
    return sg, g1


def readout_mean_linear(x, graph_ptr, weight=None, bias=None, want_pooled=False):
    """global_mean_pool over contiguous graphs + Linear head in one launch (sgx_readout_mean_linear).
    x [n, F] fp16|fp32, graph_ptr int32 [n_graphs+1], weight [C, F] fp32, bias [C] fp32."""
    _dev2d(x, "x")
    _dev(graph_ptr, "graph_ptr")
    n_graphs, F = graph_ptr.numel() - 1, x.shape[1]
    C = 0 if weight is None else weight.shape[0]
    pooled = torch.empty((n_graphs, F), dtype=torch.float32, device=x.device) if (want_pooled or weight is None) else None
    logits = torch.empty((n_graphs, C), dtype=torch.float32, device=x.device) if weight is not None else None
    w = None if weight is None else _dev(weight.detach().float().contiguous(), "weight")
    b = None if bias is None else _dev(bias.detach().float().contiguous(), "bias")
    check(lib.sgx_readout_mean_linear(dtype_code(x.dtype), n_graphs, F, C, _ptr(x), x.stride(0), _ptr(graph_ptr),
                                      _ptr(w), _ptr(b), _ptr(pooled), _ptr(logits), _stream()), "sgx_readout_mean_linear")
    if weight is None:
        return pooled
    return (logits, pooled) if want_pooled else logits


def readout_mean_backward(grad_pooled, graph_ptr, n_rows, dtype, covers_all_rows=False):
    """grad of the rows a global_mean_pool read (sgx_readout_mean_backward): grad_pooled fp32 [n_graphs, F] -> [n_rows, F]
    in `dtype`, each row its graph's gradient over the graph's size; rows of no graph get 0 (covers_all_rows: the caller
    knows there are none, e.g. graph_ptr_of(batch), and the result needs no clearing first)."""
    _dev2d(grad_pooled, "grad_pooled")
    _dev(graph_ptr, "graph_ptr")
    g = grad_pooled.float().contiguous()
    out = (torch.empty if covers_all_rows else torch.zeros)((n_rows, g.shape[1]), dtype=dtype, device=g.device)
    check(lib.sgx_readout_mean_backward(dtype_code(dtype), graph_ptr.numel() - 1, g.shape[1], _ptr(g), _ptr(graph_ptr), _ptr(out),
                                        out.stride(0), _stream()), "sgx_readout_mean_backward")
    return out


class ReadoutMean(torch.autograd.Function):
    """global_mean_pool over contiguous graphs as one launch each way (MOL cell 18's pooling inside the training step):
    forward = sgx_readout_mean_linear without a head (fp32 means in row order), backward = sgx_readout_mean_backward."""

    @staticmethod
    def forward(ctx, x, graph_ptr, covers_all_rows=False):
        ctx.save_for_backward(graph_ptr)
        ctx.about_x = (x.shape[0], x.dtype, bool(covers_all_rows))
        return readout_mean_linear(x.contiguous(), graph_ptr)

    @staticmethod
    def backward(ctx, grad):
        (graph_ptr,) = ctx.saved_tensors
        n, dtype, covers = ctx.about_x
        return readout_mean_backward(grad, graph_ptr, n, dtype, covers), None, None


def graph_ptr_of(batch):
    """graph_ptr [n_graphs + 1] int32 of a sorted PyG `batch` vector, kept on the tensor (an epoch loop passes the same
    batch every step)."""
    def build():
        counts = torch.bincount(batch)
        ptr = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=batch.device)
        ptr[1:] = torch.cumsum(counts, 0)
        return ptr
    return cached_on(batch, ("graph_ptr",), build)


def relu_mask_backward_(out, grad):
    """grad[out == 0] = 0 in place (RPYNQ.backward, MOL cell 16)."""
    _dev(out, "out")
    _dev(grad, "grad")
    check(lib.sgx_relu_mask_backward(dtype_code(out.dtype), _ptr(out), dtype_code(grad.dtype), _ptr(grad),
                                     out.numel(), _stream()), "sgx_relu_mask_backward")
    return grad
