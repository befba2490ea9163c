"""ctypes front end of the CPU oracle (oracle/sgx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under sgracex1_amd/ imports this module.

Also holds `layer_scipy`, the reference's own CPU check restated verbatim:
`csr_matrix(adj) @ (csr_matrix(fea) @ w)` (jupyter/test/mmult-master.ipynb cells
51-53) -- used to cross-check the C code, not as a second oracle.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/liborc.so with gcc (a few seconds)."""
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "sgx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s", "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_f32_to_f16.restype = ctypes.c_uint16
        _LIB.orc_f32_to_f16.argtypes = [ctypes.c_float]
        _LIB.orc_f16_to_f32.restype = ctypes.c_float
        _LIB.orc_f16_to_f32.argtypes = [ctypes.c_uint16]
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _f16bits(a):
    """float16 array (or anything castable) -> contiguous uint16 bit patterns."""
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    if a.dtype != np.float16:
        a = to_half(a)
    return a.view(np.uint16)


def to_half(a):
    """float32 -> float16 with the oracle's own RNE conversion (== numpy's)."""
    a = _f32(a)
    out = np.empty(a.shape, dtype=np.uint16)
    lib().orc_f32_to_f16_array(_p(a), _p(out), ctypes.c_int64(a.size))
    return out.view(np.float16)


def layer_f64(gemm_mode, relu, adj, fea, Wt, N=None, M_adj=None, M_fea=None, h_round=0,
              return_h=False):
    """D = relu?(A @ (X @ W)) in exact math (double accumulation).

    adj = (rowptr, col, val); fea = (rowptr, col, val) for gemm_mode 0 or a dense
    [M_adj, M_fea] array for gemm_mode 1; Wt = W transposed, [P, M_fea].
    """
    rp_a, ci_a, va_a = _i32(adj[0]), _i32(adj[1]), _f32(adj[2])
    Wt = _f32(Wt)
    P, mf = Wt.shape
    N = len(rp_a) - 1 if N is None else N
    M_adj = N if M_adj is None else M_adj
    M_fea = mf if M_fea is None else M_fea
    if gemm_mode == 0:
        rp_x, ci_x, va_x = _i32(fea[0]), _i32(fea[1]), _f32(fea[2])
    else:
        rp_x, ci_x, va_x = None, None, _f32(fea).reshape(-1)
    D = np.empty((N, P), dtype=np.float32)
    H = np.empty((M_adj, P), dtype=np.float32) if return_h else None
    rc = lib().orc_layer_f64(gemm_mode, relu, N, M_adj, M_fea, P, _p(rp_a), _p(ci_a), _p(va_a),
                             _p(rp_x), _p(ci_x), _p(va_x), _p(Wt), h_round, _p(D), _p(H))
    assert rc == 0, rc
    return (D, H) if return_h else D


def spmm_f32(relu, adj, H, P=None, rows=None):
    """D = relu?(A @ H[:, :P]); float accumulate, plain row loop (the CPU baseline kernel)."""
    rp_a, ci_a, va_a = _i32(adj[0]), _i32(adj[1]), _f32(adj[2])
    H = _f32(H)
    N = len(rp_a) - 1
    P = H.shape[1] if P is None else P
    r0, r1 = (0, N) if rows is None else rows
    D = np.zeros((N, P), dtype=np.float32)
    rc = lib().orc_spmm_f32(relu, ctypes.c_int64(r0), ctypes.c_int64(r1), P,
                            ctypes.c_int64(H.shape[1]), ctypes.c_int64(P),
                            _p(rp_a), _p(ci_a), _p(va_a), _p(H), _p(D))
    assert rc == 0, rc
    return D


def spmm_f32_into(relu, rowptr, col, val, H, D, row_begin, row_end, P):
    """In-place row-range form of spmm_f32 (no copies; callable from several threads)."""
    rc = lib().orc_spmm_f32(relu, ctypes.c_int64(row_begin), ctypes.c_int64(row_end), P,
                            ctypes.c_int64(H.shape[1]), ctypes.c_int64(D.shape[1]),
                            _p(rowptr), _p(col), _p(val), _p(H), _p(D))
    assert rc == 0, rc


def xw_dense_f32_into(X, W, H, row_begin, row_end):
    """H[rows] = X[rows] @ W, W row-major [M, P]; in place, thread-safe over disjoint rows."""
    M, P = W.shape
    rc = lib().orc_xw_dense_f32(ctypes.c_int64(row_begin), ctypes.c_int64(row_end), M, P,
                                ctypes.c_int64(X.shape[1]), ctypes.c_int64(H.shape[1]), _p(X), _p(W), _p(H))
    assert rc == 0, rc


def layer_refhalf(gemm_mode, relu, adj, fea, Wt, N=None, M_adj=None, spmm_block=1, lat_fea=4,
                  lat_adj=4, fea_threads=1, adj_threads=1, return_h=False):
    """Bit-accurate model of the reference HALF build; returns float16 arrays."""
    rp_a, ci_a, va_a = _i32(adj[0]), _i32(adj[1]), _f16bits(adj[2])
    Wt = np.ascontiguousarray(Wt)
    P, M_fea = Wt.shape
    Wt = _f16bits(Wt)
    N = len(rp_a) - 1 if N is None else N
    M_adj = N if M_adj is None else M_adj
    if gemm_mode == 0:
        rp_x, ci_x, va_x = _i32(fea[0]), _i32(fea[1]), _f16bits(fea[2])
    else:
        rp_x, ci_x, va_x = None, None, _f16bits(np.asarray(fea).reshape(-1))
    D = np.empty((N, P), dtype=np.uint16)
    H = np.empty((M_adj, P), dtype=np.uint16) if return_h else None
    rc = lib().orc_layer_refhalf(gemm_mode, relu, N, M_adj, M_fea, P, _p(rp_a), _p(ci_a), _p(va_a),
                                 _p(rp_x), _p(ci_x), _p(va_x), _p(Wt), spmm_block, lat_fea, lat_adj,
                                 fea_threads, adj_threads, _p(D), _p(H))
    assert rc == 0, rc
    D = D.view(np.float16)
    return (D, H.view(np.float16)) if return_h else D


def gat_f64(relu, adj, Wh, att, alpha=0.2):
    """Single-head GAT forward on the stored edges; returns (D, E, S)."""
    rp_a, ci_a, va_a = _i32(adj[0]), _i32(adj[1]), _f32(adj[2])
    Wh = _f32(Wh)
    att = _f32(att).reshape(-1)
    N, F = Wh.shape
    assert att.size == 2 * F
    nnz = len(ci_a)
    D = np.empty((N, F), dtype=np.float32)
    E = np.empty(nnz, dtype=np.float32)
    S = np.empty(nnz, dtype=np.float32)
    rc = lib().orc_gat_f64(relu, N, F, ctypes.c_float(alpha), _p(rp_a), _p(ci_a), _p(va_a),
                           _p(Wh), _p(att), _p(D), _p(E), _p(S))
    assert rc == 0, rc
    return D, E, S


def layer_scipy(gemm_mode, relu, adj, fea, W, N=None, M_fea=None):
    """The reference's own software check (mmult-master.ipynb cells 51-53), float32."""
    from scipy.sparse import csr_matrix
    W = np.asarray(W, dtype=np.float32)          # [M_fea, P]
    N = len(adj[0]) - 1 if N is None else N
    A = csr_matrix((np.asarray(adj[2], np.float32), adj[1], adj[0]), shape=(N, N))
    if gemm_mode == 0:
        X = csr_matrix((np.asarray(fea[2], np.float32), fea[1], fea[0]), shape=(N, W.shape[0]))
        out = A @ (X @ W)
    else:
        out = A @ (np.asarray(fea, np.float32).reshape(N, -1) @ W)
    out = np.asarray(out, dtype=np.float32)
    return np.maximum(out, 0) if relu else out
