#!/usr/bin/env python3
"""Dense X.W with K > 128: the kernel that keeps W^T in LDS (xw_dense_wlds.hip) against the 128 x 128 tile kernel it
replaces, on the Reddit shape and three others.  The output allocation (torch caching allocator) is inside both timings.

    python tools/xw_dense_long_k_probe.py > gpurun_out/xw_dense_long_k.jsonl
"""
import os, sys, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgracex1_amd import ops
from sgracex1_amd import _lib
g = torch.Generator(device="cuda"); g.manual_seed(1)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (n, M, P) in [(232965, 602, 128), (232965, 602, 256), (2449029, 300, 128), (1000000, 1000, 64)]:
    X = (torch.rand((n, M), generator=g, device="cuda") - 0.4).half()
    Wt = ((torch.rand((P, M), generator=g, device="cuda") * 2 - 1) / M ** 0.5).half()
    new = t(lambda: ops.xw_dense(X, Wt))
    os.environ["SGX_XW_NO_WLDS"] = "1"
    _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
    old = t(lambda: ops.xw_dense(X, Wt))
    del os.environ["SGX_XW_NO_WLDS"]
    _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
    b = n * M * 2 + n * P * 2
    print(json.dumps({"rows": n, "M": M, "P": P, "ms_w_in_lds": round(new, 4), "ms_tile_kernel": round(old, 4), "GBps_new": round(b / new / 1e6, 1), "frac_8TBps": round(b / new / 1e6 / 8000, 3)}), flush=True)
    del X, Wt
