#!/usr/bin/env python3
"""Times the sparse X.W stage (sgx_xw_sparse) on the bench workload's feature matrix: the LDS-resident weight
slice against the gather kernel (SGX_XW_SPARSE_NO_LDS), and the gather kernel with the lanes-per-row split forced
(SGX_SPMM_CPL = 1, 2, 4)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sgracex1_amd import graphs, ops  # noqa: E402
from sgracex1_amd import _lib
from sgracex1_amd._lib import check, lib  # noqa: E402
from sgracex1_amd.hipevents import Event  # noqa: E402

dev = torch.device("cuda")
wl = dict(bench.WORKLOADS["s100m"])
wl["edges"] = 1_000_000                       # the adjacency is not used here
_A, X, W1t, _W2t = bench.make_inputs(torch, graphs, ops, wl, 0, 1, dev)
X.plan
W = ops.transpose(W1t)                        # [F_in, hidden]
H = torch.empty((X.n_rows, W.shape[1]), dtype=torch.float16, device=dev)
stream_id = torch.cuda.current_stream().cuda_stream
stream = ctypes.c_void_p(stream_id)


def run():
    check(lib.sgx_xw_sparse(0, 0, 1, X.n_rows, X.n_cols, W.shape[1], X.rowptr.data_ptr(), X.col.data_ptr(), X.val.data_ptr(),
                            W.data_ptr(), W.stride(0), H.data_ptr(), H.stride(0), X.plan.handle, None, 0, stream), "xw")


def timed(iters=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        b, e = Event(), Event()
        b.record(stream_id)
        run()
        e.record(stream_id)
        ts.append(b.elapsed_ms(e))
    return min(ts), sum(ts) / len(ts)


rec = {"rows": X.n_rows, "nnz": X.nnz, "f_in": X.n_cols, "P": W.shape[1]}
if os.environ.get("SGX_PROBE_ONLY"):                    # under rocprofv3 --pmc: a few launches of one form
    if os.environ["SGX_PROBE_ONLY"] == "gather":
        os.environ["SGX_XW_SPARSE_NO_LDS"] = "1"
        _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    print(json.dumps(rec), flush=True)
    sys.exit(0)
if os.environ.get("SGX_PROBE_QUICK"):                   # tools/sweep_xw_sparse.py: the LDS form only, and that it equals the gather form
    rec["ms_lds"] = [[round(v, 4) for v in timed()] for _ in range(2)]
    run()
    H_lds = H.clone()
    os.environ["SGX_XW_SPARSE_NO_LDS"] = "1"
    _lib.lib.sgx_reload_env()
    run()
    rec["lds_equal_to_gather"] = bool(torch.equal(H_lds, H))
    print(json.dumps(rec), flush=True)
    sys.exit(0)
# the LDS-resident weight slice (default for a matrix this size) against the gather kernel, interleaved in one process
for rnd in range(3):
    rec.setdefault("ms_lds", []).append([round(v, 4) for v in timed()])
    os.environ["SGX_XW_SPARSE_NO_LDS"] = "1"
    _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
    rec.setdefault("ms_gather", []).append([round(v, 4) for v in timed()])
    del os.environ["SGX_XW_SPARSE_NO_LDS"]
    _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
run()
H_lds = H.clone()
os.environ["SGX_XW_SPARSE_NO_LDS"] = "1"
_lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
for cpl in (1, 2, 4):
    os.environ["SGX_SPMM_CPL"] = str(cpl)
    _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
    rec[f"ms_cpl{cpl}"] = [round(v, 4) for v in timed()]
del os.environ["SGX_SPMM_CPL"]
_lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
rec["ms_policy"] = [round(v, 4) for v in timed()]
ref = ops.spmm(X, W, relu=False)              # the A.H entry point on the same operands: same sums, same bits
run()
rec["equal_to_agg_entry"] = bool(torch.equal(ref, H))
rec["lds_equal_to_gather"] = bool(torch.equal(H_lds, H))
del os.environ["SGX_XW_SPARSE_NO_LDS"]
_lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
print(json.dumps(rec), flush=True)
