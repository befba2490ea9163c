#!/usr/bin/env python3
"""Two questions about the sparse X.W stage with W in LDS, answered from the host side only (same kernel, other inputs):
what do LDS bank conflicts cost (random column indices against conflict-free patterns), and what would a second workgroup
per CU buy (a W of half the rows, whose slice leaves room for two workgroups, at the same entries per row).  One JSON line."""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from sgracex1_amd import ops
from sgracex1_amd.hipevents import Event
dev = torch.device("cuda")
n, f_in, P = 1 << 22, 1433, 64
g = torch.Generator(device=dev); g.manual_seed(1)
nnz_x = int(n * f_in * 0.0127)
xr = torch.randint(0, n, (nnz_x,), generator=g, device=dev, dtype=torch.int64)
xc = torch.randint(0, f_in, (nnz_x,), generator=g, device=dev, dtype=torch.int64)
key = torch.unique(xr * f_in + xc)
xr = torch.div(key, f_in, rounding_mode="floor"); xc = key - xr * f_in
xp = torch.zeros(n + 1, dtype=torch.int64, device=dev); torch.cumsum(torch.bincount(xr, minlength=n), 0, out=xp[1:])
val = torch.ones(key.numel(), device=dev).half()
W = ((torch.rand((f_in, P), generator=g, device=dev) * 2 - 1) / 8).half()
H = torch.empty((n, P), dtype=torch.float16, device=dev)
def timed(X):
    X.plan
    for _ in range(3): ops.xw_sparse(X, W, out=H)
    torch.cuda.synchronize(); s = torch.cuda.current_stream().cuda_stream; ts = []
    for _ in range(10):
        b, e = Event(), Event(); b.record(s); ops.xw_sparse(X, W, out=H); e.record(s); ts.append(b.elapsed_ms(e))
    return round(min(ts), 4)
rp = xp.to(torch.int32)
res = {"random_cols": timed(ops.Csr(rp, xc.to(torch.int32), val, f_in))}
res["all_cols_zero"] = timed(ops.Csr(rp, torch.zeros_like(xc, dtype=torch.int32), val, f_in))
seq = (torch.arange(key.numel(), device=dev) % f_in).to(torch.int32)
res["sequential_cols"] = timed(ops.Csr(rp, seq, val, f_in))
res["cols_multiple_of_4"] = timed(ops.Csr(rp, (xc // 4 * 4).to(torch.int32), val, f_in))
perm = (xc * 4 % f_in).to(torch.int32)
res["cols_times4_mod"] = timed(ops.Csr(rp, perm, val, f_in))
# the same entries against a W of 700 rows: its 32-column slice is 45 KB, two workgroups (32 wavefronts) fit a CU
f2 = 700
W = ((torch.rand((f2, P), generator=g, device=dev) * 2 - 1) / 8).half()
res["f_in_700_two_workgroups_per_cu_random_cols"] = timed(ops.Csr(rp, (xc % f2).to(torch.int32), val, f2))
f3 = 1200      # one workgroup per CU again (77 KB slices... two fit 160 KB? 2 x 77 = 154: yes) 
W = ((torch.rand((f3, P), generator=g, device=dev) * 2 - 1) / 8).half()
res["f_in_1200_random_cols"] = timed(ops.Csr(rp, (xc % f3).to(torch.int32), val, f3))
f4 = 1300      # 83 KB slices: one workgroup per CU
W = ((torch.rand((f4, P), generator=g, device=dev) * 2 - 1) / 8).half()
res["f_in_1300_random_cols"] = timed(ops.Csr(rp, (xc % f4).to(torch.int32), val, f4))
# the Cora shape itself with narrower slices: 4 slices of 16 columns (2 lanes per row) are 46 KB each, two workgroups per CU
from sgracex1_amd import _lib
W = ((torch.rand((f_in, P), generator=g, device=dev) * 2 - 1) / 8).half()
X = ops.Csr(rp, xc.to(torch.int32), val, f_in)
ref = ops.xw_sparse(X, W).clone()
res["f_in_1433_lpr4_again"] = timed(X)
with _lib.tuning(SGX_XW_SPARSE_LPR="2"):
    res["f_in_1433_lpr2_four_slices_two_workgroups_per_cu"] = timed(X)
    res["lpr2_same_bits"] = bool(torch.equal(ops.xw_sparse(X, W), ref))
print(json.dumps(res))
