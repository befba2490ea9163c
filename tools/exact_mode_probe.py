#!/usr/bin/env python3
"""SGX_ACC_REF_HALF (the reference's half arithmetic, bit for bit) timed on the bench graph beside the
default fp32-accumulate aggregation."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sgracex1_amd import graphs, ops  # noqa: E402
from sgracex1_amd.hipevents import Event  # noqa: E402

dev = torch.device("cuda")
wl = bench.WORKLOADS["s100m"]
A, X, W1t, _W2t = bench.make_inputs(torch, graphs, ops, wl, 0, 1, dev)
n, P = A.n_rows, wl["hidden"]
H = torch.rand((n, P), device=dev).half()
A.plan
s = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        b, e = Event(), Event()
        b.record(s)
        fn()
        e.record(s)
        ts.append(b.elapsed_ms(e))
    return round(min(ts), 3)


rec = {"edges": A.nnz, "P": P,
       "default_ms": timed(lambda: ops.spmm(A, H, relu=True)),
       "exact_spmm_block1_ms": timed(lambda: ops.spmm(A, H, relu=True, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=1)),
       "exact_spmm_block4_ms": timed(lambda: ops.spmm(A, H, relu=True, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4)),
       "exact_sparse_xw_ms": timed(lambda: ops.spmm(X, ops.transpose(W1t), relu=False, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4))}
d0 = ops.spmm(A, H, relu=True).float()
d1 = ops.spmm(A, H, relu=True, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4).float()
rec["max_abs_diff_exact_vs_default"] = float((d0 - d1).abs().max())
print(json.dumps(rec), flush=True)
