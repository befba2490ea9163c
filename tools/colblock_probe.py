#!/usr/bin/env python3
"""Does cutting the gathered table into column blocks that fit the Infinity Cache pay?  The
aggregation of the bench graph in k passes (pass b takes the edges whose column lies in block b of
the table and accumulates into fp32 partials, sgx_spmm_csr_acc) against the single pass."""
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
A, _X, _W1t, _W2t = bench.make_inputs(torch, graphs, ops, wl, 0, 1, dev)
n, P = A.n_rows, wl["hidden"]
H = torch.rand((n, P), device=dev).half()
A.plan
D = torch.empty((n, P), dtype=torch.float16, device=dev)
s = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        b, e = Event(), Event()
        b.record(s)
        fn()
        e.record(s)
        ts.append(b.elapsed_ms(e))
    return round(min(ts), 4), round(sum(ts) / len(ts), 4)


rec = {"single_pass_ms": timed(lambda: ops.spmm(A, H, relu=True, out=D))}
ref = ops.spmm(A, H, relu=True).clone()
deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
for k in (2, 4):
    bounds = [n * b // k for b in range(k + 1)]
    parts = []
    for b in range(k):
        m = (A.col >= bounds[b]) & (A.col < bounds[b + 1])
        rp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        rp[1:] = torch.cumsum(torch.bincount(row[m], minlength=n), 0)
        blk = ops.Csr(rp, (A.col[m] - bounds[b]).contiguous(), A.val[m].contiguous(), bounds[b + 1] - bounds[b])
        blk.plan
        parts.append((blk, H[bounds[b]:bounds[b + 1]]))
    acc = torch.empty((n, P), dtype=torch.float32, device=dev)

    def run():
        part = ops.spmm_acc(parts[0][0], parts[0][1], partial_out=True)
        for blk, tab in parts[1:-1]:
            part = ops.spmm_acc(blk, tab, acc_in=part, partial_out=True)
        return ops.spmm_acc(parts[-1][0], parts[-1][1], relu=True, acc_in=part, out=D)

    rec[f"{k}_column_blocks_ms"] = timed(run)
    rec[f"{k}_blocks_equal_fraction"] = float((run() == ref).float().mean())
    del parts
print(json.dumps(rec), flush=True)
