#!/usr/bin/env python3
"""The one-step tail of a degree-ordered plan (rows of at most 8 edges, 64 to a wavefront: spmm_short_rows) against the
sblock path for the same rows (SGX_SPMM_NO_SHORT_TAIL), interleaved in one process, on the power-law S-100M aggregation
and on the sub-matrices of its short rows.  Checks the two give the same bits."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgracex1_amd import _lib, graphs, ops  # noqa: E402
from sgracex1_amd.hipevents import Event  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
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
    ts.sort()
    return round(ts[0], 4), round(ts[len(ts) // 2], 4)


def main():
    dev = torch.device("cuda")
    scale, edges = 22, 100_000_000
    n = 1 << scale
    A = graphs.rmat_graph(scale, edges, seed=12345, device=dev)
    A.plan
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    H = torch.rand((n, 64), generator=g, device=dev).half()
    D = torch.empty((n, 64), dtype=torch.float16, device=dev)
    rec = {"graph": "rmat 2^22 / 100 M", "edges": A.nnz, "reordered": A.plan.reordered}
    for rnd in range(3):
        rec.setdefault("ms_short_tail", []).append(timed(lambda: ops.spmm(A, H, relu=True, out=D)))
        new = D.clone()
        with _lib.tuning(SGX_SPMM_NO_SHORT_TAIL="1"):
            rec.setdefault("ms_sblock_tail", []).append(timed(lambda: ops.spmm(A, H, relu=True, out=D)))
        rec["same_bits"] = bool(torch.equal(new, D))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
