#!/usr/bin/env python3
"""Times the aggregation kernel with the lanes-per-row split forced (SGX_SPMM_CPL = 1, 2, 4) on
graphs of different mean degree, to set the policy in choose_cpl (csrc/spmm_csr.hip)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sgracex1_amd import graphs, ops  # noqa: E402
from sgracex1_amd import _lib
from sgracex1_amd.hipevents import Event  # noqa: E402


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for _ in range(iters):
        b, e = Event(), Event()
        b.record(s)
        fn()
        e.record(s)
        ts.append(b.elapsed_ms(e))
    return min(ts)


def probe(name, A, table, out):
    A.plan
    rec = {"case": name, "rows": A.n_rows, "nnz": A.nnz, "avg_deg": round(A.nnz / A.n_rows, 2), "P": table.shape[1]}
    for cpl in (1, 2, 4):
        os.environ["SGX_SPMM_CPL"] = str(cpl)
        _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
        rec[f"ms_cpl{cpl}"] = round(timed(lambda: ops.spmm(A, table, relu=False, out=out)), 4)
    del os.environ["SGX_SPMM_CPL"]
    _lib.lib.sgx_reload_env()      # the library reads its overrides once; have it read them again
    rec["ms_policy"] = round(timed(lambda: ops.spmm(A, table, relu=False, out=out)), 4)
    print(json.dumps(rec), flush=True)


dev = "cuda"
n = 1 << 22
H = torch.rand((n, 64), device=dev).half()
D = torch.empty((n, 64), device=dev, dtype=torch.float16)
for edges in (100_000_000, 50_000_000, 25_000_000, 12_000_000, 4_000_000):
    A = graphs.uniform_graph(n, edges, seed=1)
    probe(f"uniform E={edges // 1_000_000}M", A, H, D)
    del A
A = graphs.rmat_graph(22, 100_000_000)
probe("rmat 100M", A, H, D)
del A
# the sparse feature matrix of bench.py (X . W with W as the table)
g = torch.Generator(device=dev)
g.manual_seed(5)
f_in = 1433
nnz_x = int(n * f_in * 0.0127)
key = torch.unique(torch.randint(0, n, (nnz_x,), generator=g, device=dev) * f_in +
                   torch.randint(0, f_in, (nnz_x,), generator=g, device=dev))
xr = torch.div(key, f_in, rounding_mode="floor")
X = ops.Csr.from_coo(xr.to(torch.int32), (key - xr * f_in).to(torch.int32), torch.ones(key.numel(), device=dev).half(), n, f_in)
W = torch.rand((f_in, 64), device=dev).half()
probe("sparse X.W (1433 -> 64)", X, W, D)
H128 = torch.rand((n // 4, 128), device=dev).half()
D128 = torch.empty((n // 4, 128), device=dev, dtype=torch.float16)
A = graphs.uniform_graph(n // 4, 30_000_000, seed=2)
probe("uniform N=1M E=30M P=128", A, H128, D128)
