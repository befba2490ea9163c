#!/usr/bin/env python3
"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in
THIS path's access pattern (MI355X_MICROARCH.md "HBM": FETCH_SIZE is only calibrated for wide
streaming reads).  Run under  rocprofv3 --pmc FETCH_SIZE ...  and  --pmc WRITE_SIZE ...

  launch A  spmm_kernel on a permutation graph: N = 2^24 rows, one edge per row to a
            distinct random row of H [N, 64] fp16 (2 GiB, far beyond L2 + Infinity Cache), so
            every 128-byte row of H is gathered exactly once:
              reads  = N*128 (H) + N*(4+2) (col,val) + (N+1)*4 (rowptr)   writes = N*128
  launch B  a 2 GiB device-to-device copy (torch), the streaming pattern of the guide.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sgracex1_amd import ops  # noqa: E402

n = 1 << 24
dev = torch.device("cuda")
g = torch.Generator(device=dev)
g.manual_seed(1)
perm = torch.randperm(n, generator=g, device=dev).to(torch.int32)
rowptr = torch.arange(n + 1, device=dev, dtype=torch.int32)
A = ops.Csr(rowptr, perm, torch.ones(n, device=dev, dtype=torch.float16), n)
H = torch.rand((n, 64), device=dev).half()
out = torch.empty_like(H)
for _ in range(3):
    ops.spmm(A, H, relu=False, out=out, use_plan=False)
torch.cuda.synchronize()
assert torch.equal(out, H[perm.long()])
x = torch.empty(1 << 30, dtype=torch.float16, device=dev).normal_()
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
print("gather_read_bytes", n * 128 + n * 6 + (n + 1) * 4, "gather_write_bytes", n * 128,
      "copy_bytes_each_way", x.numel() * 2)
