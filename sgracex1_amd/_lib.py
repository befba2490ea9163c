"""ctypes binding of libsgx.so (the C ABI declared in include/sgx.h).

The library is the only backend.  If it is missing or does not load, importing this module
raises -- there is no CPU or PyTorch fallback for the accelerated path.
"""
import ctypes
import os

# PyTorch-ROCm ships its own HIP runtime; importing it first makes that copy the one this process
# (and libsgx.so, whose libamdhip64 dependency then resolves to it) uses -- two runtimes in one
# process cannot share streams or allocations.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGX_LIB_PATH selects another build of the same library (tools/sweep_spmm.py compares variants)
LIB_PATH = os.environ.get("SGX_LIB_PATH") or os.path.join(_HERE, "csrc", "libsgx.so")

SGX_F16, SGX_F32 = 0, 1
SGX_ACC_F32, SGX_ACC_REF_HALF = 0, 1
SGX_ORDER_REFERENCE, SGX_ORDER_AGGREGATE_FIRST = 0, 1      # sgx_layer_order
SGX_QUANT_INT8 = 2                                          # sgx_quant.flags: integer operands on the int8 matrix cores
SGX_QUANT_INT8_AUTO = 4                                     # ... where they are the faster form (M_fea > 128)

# every symbol include/sgx.h declares (tests/test_abi.py checks header and library against this)
SYMBOLS = [
    "sgx_plan_create", "sgx_plan_create_ex", "sgx_plan_destroy", "sgx_plan_long_rows", "sgx_plan_long_threshold", "sgx_plan_natural_utilization",
    "sgx_plan_reordered", "sgx_plan_export",
    "sgx_fake_quantize", "sgx_requantize",
    "sgx_layer_workspace_bytes", "sgx_layer_forward",
    "sgx_spmm_csr", "sgx_spmm_csr_acc", "sgx_spmm_scratch_bytes", "sgx_xw_dense", "sgx_xw_sparse", "sgx_transpose",
    "sgx_gat_aggregate", "sgx_gat_scratch_bytes", "sgx_csr_validate", "sgx_coo_to_csr", "sgx_relu_mask_backward",
    "sgx_xt_g", "sgx_xt_g_workspace_bytes", "sgx_readout_mean_linear", "sgx_readout_mean_backward", "sgx_gat_backward_edges",
    "sgx_stream_copy", "sgx_xw_dense_act", "sgx_event_create", "sgx_event_destroy", "sgx_event_record", "sgx_event_elapsed_ms",
    "sgx_gat_aggregate_fill", "sgx_col_sums", "sgx_col_sums_scratch_bytes", "sgx_pack_rows",
    "sgx_code_bias", "sgx_quantize_codes_i8", "sgx_xw_dense_i8", "sgx_xw_dense_i8_workspace_bytes",
    "sgx_version", "sgx_status_string", "sgx_reload_env",
]


class SgxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        super().__init__(f"{where}: {status_string(status)} (sgx_status {status})")


SGX_QUANT_ADJ_DONE = 1


class Quant(ctypes.Structure):
    """struct sgx_quant -- field order and types must match include/sgx.h."""
    _fields_ = [
        ("qbits", ctypes.c_int32), ("scale_fea", ctypes.c_int32), ("internal_bits", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("inv_scale_fea", ctypes.c_float), ("zero_fea", ctypes.c_float),
        ("inv_scale_w", ctypes.c_float), ("zero_w", ctypes.c_float),
        ("inv_scale_adj", ctypes.c_float), ("zero_adj", ctypes.c_float),
        ("deq_factor", ctypes.c_float), ("reserved", ctypes.c_float),
        ("nnz_adj", ctypes.c_int64), ("nnz_fea", ctypes.c_int64),
    ]


class LayerDesc(ctypes.Structure):
    """struct sgx_layer_desc -- field order and types must match include/sgx.h."""
    _fields_ = [
        ("gemm_mode", ctypes.c_int32), ("relu", ctypes.c_int32), ("gat_mode", ctypes.c_int32),
        ("N_adj", ctypes.c_int32), ("M_adj", ctypes.c_int32), ("M_fea", ctypes.c_int32),
        ("P_w", ctypes.c_int32), ("bias_count", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("acc_mode", ctypes.c_int32), ("spmm_block", ctypes.c_int32), ("gat_fill_dead_rows", ctypes.c_int32),
        ("B", ctypes.c_void_p), ("D", ctypes.c_void_p),
        ("rowPtr_fea", ctypes.c_void_p), ("columnIndex_fea", ctypes.c_void_p), ("values_fea", ctypes.c_void_p),
        ("rowPtr_adj", ctypes.c_void_p), ("columnIndex_adj", ctypes.c_void_p), ("values_adj", ctypes.c_void_p),
        ("attention", ctypes.c_void_p), ("E", ctypes.c_void_p), ("S", ctypes.c_void_p),
        ("alpha", ctypes.c_float), ("fea_threads", ctypes.c_int32), ("adj_threads", ctypes.c_int32),
        ("gat_heads", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("plan_adj", ctypes.c_void_p), ("plan_fea", ctypes.c_void_p),
        ("ev_agg_begin", ctypes.c_void_p), ("ev_agg_end", ctypes.c_void_p),
        ("quant", ctypes.POINTER(Quant)),
        ("order", ctypes.c_int32),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first (python -m sgracex1_amd.build); "
            "sgracex1_amd has no other backend")
    lib = ctypes.CDLL(LIB_PATH)
    c_int, c_i64, vp, sz = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    lib.sgx_plan_create.argtypes = [ctypes.POINTER(vp), vp, c_int, c_int, vp]
    lib.sgx_plan_create.restype = c_int
    lib.sgx_plan_create_ex.argtypes = [ctypes.POINTER(vp), vp, c_int, c_int, c_int, vp]
    lib.sgx_plan_create_ex.restype = c_int
    lib.sgx_plan_destroy.argtypes = [vp]
    lib.sgx_plan_destroy.restype = None
    lib.sgx_plan_long_rows.argtypes = [vp]
    lib.sgx_plan_long_rows.restype = c_int
    lib.sgx_plan_long_threshold.argtypes = [vp]
    lib.sgx_plan_long_threshold.restype = c_int
    lib.sgx_plan_natural_utilization.argtypes = [vp]
    lib.sgx_plan_natural_utilization.restype = ctypes.c_float
    lib.sgx_plan_reordered.argtypes = [vp]
    lib.sgx_plan_reordered.restype = c_int
    lib.sgx_plan_export.argtypes = [vp, c_int, vp, ctypes.c_int64, vp]
    lib.sgx_plan_export.restype = ctypes.c_int64
    lib.sgx_fake_quantize.argtypes = [c_int, c_int, ctypes.c_float, ctypes.c_float, c_i64, vp, vp, vp]
    lib.sgx_fake_quantize.restype = c_int
    lib.sgx_requantize.argtypes = [c_int, c_int, c_i64, vp, c_int, c_int, vp]
    lib.sgx_requantize.restype = c_int
    lib.sgx_layer_workspace_bytes.argtypes = [ctypes.POINTER(LayerDesc)]
    lib.sgx_layer_workspace_bytes.restype = sz
    lib.sgx_layer_forward.argtypes = [ctypes.POINTER(LayerDesc), vp]
    lib.sgx_layer_forward.restype = c_int
    lib.sgx_spmm_csr.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_i64, vp, c_i64,
                                 vp, vp, sz, vp]
    lib.sgx_spmm_csr.restype = c_int
    lib.sgx_spmm_csr_acc.argtypes = [c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_i64, vp, c_i64,
                                     vp, vp, c_i64, vp, vp, sz, vp]
    lib.sgx_spmm_csr_acc.restype = c_int
    lib.sgx_spmm_scratch_bytes.argtypes = [vp, c_int]
    lib.sgx_spmm_scratch_bytes.restype = sz
    lib.sgx_xw_dense.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, vp, c_i64, vp, c_i64, vp, c_i64, vp]
    lib.sgx_xw_dense.restype = c_int
    lib.sgx_xw_sparse.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, c_i64, vp, c_i64,
                                  vp, vp, sz, vp]
    lib.sgx_xw_sparse.restype = c_int
    lib.sgx_transpose.argtypes = [c_int, c_int, c_int, vp, c_i64, vp, c_i64, vp]
    lib.sgx_transpose.restype = c_int
    lib.sgx_gat_scratch_bytes.argtypes = [c_int, c_int, c_int, c_int, vp]
    lib.sgx_gat_scratch_bytes.restype = sz
    lib.sgx_gat_aggregate.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_float, vp, vp, vp, vp, c_i64, vp,
                                      vp, c_i64, vp, vp, vp, vp, vp]
    lib.sgx_gat_aggregate.restype = c_int
    lib.sgx_gat_aggregate_fill.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_float, vp, vp, vp, vp, c_i64, vp,
                                           vp, c_i64, vp, vp, vp, c_i64, vp, vp, vp]
    lib.sgx_gat_aggregate_fill.restype = c_int
    lib.sgx_col_sums_scratch_bytes.argtypes = [c_int]
    lib.sgx_col_sums_scratch_bytes.restype = sz
    lib.sgx_col_sums.argtypes = [c_int, c_int, c_int, vp, c_i64, vp, vp, vp]
    lib.sgx_col_sums.restype = c_int
    lib.sgx_code_bias.argtypes = [c_int, c_int]
    lib.sgx_code_bias.restype = c_int
    lib.sgx_quantize_codes_i8.argtypes = [c_int, c_int, ctypes.c_float, ctypes.c_float, c_int, c_int, vp, c_i64, vp, c_i64, vp]
    lib.sgx_quantize_codes_i8.restype = c_int
    lib.sgx_xw_dense_i8_workspace_bytes.argtypes = [c_int]
    lib.sgx_xw_dense_i8_workspace_bytes.restype = sz
    lib.sgx_xw_dense_i8.argtypes = [c_int, c_int, c_int, c_int, vp, c_i64, vp, c_i64, c_int, c_int, vp, c_i64, vp, vp]
    lib.sgx_xw_dense_i8.restype = c_int
    lib.sgx_pack_rows.argtypes = [c_int, c_i64, c_int, vp, c_i64, vp, vp, c_i64, vp]
    lib.sgx_pack_rows.restype = c_int
    lib.sgx_csr_validate.argtypes = [vp, vp, c_int, c_int, c_i64, vp]
    lib.sgx_csr_validate.restype = c_int
    lib.sgx_coo_to_csr.argtypes = [vp, c_i64, c_int, vp, vp]
    lib.sgx_coo_to_csr.restype = c_int
    lib.sgx_relu_mask_backward.argtypes = [c_int, vp, c_int, vp, c_i64, vp]
    lib.sgx_relu_mask_backward.restype = c_int
    lib.sgx_gat_backward_edges.argtypes = [c_int, c_int, c_int, c_int, ctypes.c_float, vp, vp, vp, vp, vp, vp, c_i64, vp,
                                           c_i64, vp, vp, vp]
    lib.sgx_gat_backward_edges.restype = c_int
    lib.sgx_readout_mean_linear.argtypes = [c_int, c_int, c_int, c_int, vp, c_i64, vp, vp, vp, vp, vp, vp]
    lib.sgx_readout_mean_linear.restype = c_int
    lib.sgx_readout_mean_backward.argtypes = [c_int, c_int, c_int, vp, vp, vp, c_i64, vp]
    lib.sgx_readout_mean_backward.restype = c_int
    lib.sgx_xt_g_workspace_bytes.argtypes = [c_int, c_int, c_int]
    lib.sgx_xt_g_workspace_bytes.restype = sz
    lib.sgx_xt_g.argtypes = [c_int, c_int, c_int, c_int, vp, c_i64, vp, c_i64, vp, c_i64, vp, sz, vp]
    lib.sgx_xt_g.restype = c_int
    lib.sgx_xw_dense_act.argtypes = [c_int, c_int, c_int, c_int, c_int, vp, c_i64, vp, c_i64, vp, c_i64, vp]
    lib.sgx_xw_dense_act.restype = c_int
    lib.sgx_stream_copy.argtypes = [vp, vp, c_i64, vp]
    lib.sgx_stream_copy.restype = c_int
    lib.sgx_event_create.argtypes = [ctypes.POINTER(vp)]
    lib.sgx_event_create.restype = c_int
    lib.sgx_event_destroy.argtypes = [vp]
    lib.sgx_event_destroy.restype = c_int
    lib.sgx_event_record.argtypes = [vp, vp]
    lib.sgx_event_record.restype = c_int
    lib.sgx_event_elapsed_ms.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.sgx_event_elapsed_ms.restype = c_int
    lib.sgx_version.argtypes = []
    lib.sgx_version.restype = c_int
    lib.sgx_status_string.argtypes = [c_int]
    lib.sgx_status_string.restype = ctypes.c_char_p
    lib.sgx_reload_env.argtypes = []
    lib.sgx_reload_env.restype = None
    return lib


lib = _load()


def status_string(status):
    return lib.sgx_status_string(int(status)).decode()


def check(status, where):
    if status != 0:
        raise SgxError(status, where)


import contextlib


@contextlib.contextmanager
def tuning(**overrides):
    """Run a block under SGX_* tuning overrides: `with tuning(SGX_XW_NO_WLDS="1"): ...`.  The library reads its
    overrides from the environment once per process (include/sgx.h, sgx_reload_env), so changing os.environ alone does
    nothing after the first call; this sets the variables, has the library read them again, and undoes both on the way
    out.  A value of None removes the variable for the block.  For tests and probes that compare two forms of a kernel."""
    saved = {k: os.environ.get(k) for k in overrides}
    try:
        for k, v in overrides.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        lib.sgx_reload_env()
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.sgx_reload_env()
