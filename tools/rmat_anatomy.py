#!/usr/bin/env python3
"""Where the power-law S-100M aggregation spends its time: the rows of the R-MAT graph by degree class, each class as a
CSR of its own (same columns, same table) through the same sgx_spmm_csr, beside the whole graph and the uniform graph.

    python3 tools/rmat_anatomy.py [--scale 22] [--edges 100000000] > profiles/r03_rmat_anatomy.jsonl
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgracex1_amd import graphs, ops  # noqa: E402
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
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def sub_csr(A, keep_rows):
    """rows `keep_rows` (ascending ids) of A as a CSR of their own"""
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    d = deg[keep_rows]
    rp = torch.zeros(keep_rows.numel() + 1, dtype=torch.int64, device=A.col.device)
    torch.cumsum(d, 0, out=rp[1:])
    start = A.rowptr[:-1].long()[keep_rows]
    # edge ids of the kept rows, in order
    idx = torch.repeat_interleave(start - rp[:-1], d) + torch.arange(int(rp[-1]), device=A.col.device)
    return ops.Csr(rp.to(torch.int32), A.col[idx].contiguous(), A.val[idx].contiguous(), A.n_cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edges", type=int, default=100_000_000)
    ap.add_argument("--hidden", type=int, default=64)
    args = ap.parse_args()
    n = 1 << args.scale
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    H = torch.rand((n, args.hidden), generator=g, device=dev).half()
    for gen in ("uniform", "rmat"):
        A = (graphs.rmat_graph(args.scale, args.edges, seed=12345, device=dev) if gen == "rmat"
             else graphs.uniform_graph(n, args.edges, seed=12345, device=dev))
        A.plan
        D = torch.empty((n, args.hidden), dtype=torch.float16, device=dev)
        mn, med = timed(lambda: ops.spmm(A, H, relu=True, out=D))
        deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
        print(json.dumps({"graph": gen, "class": "all", "rows": n, "edges": A.nnz, "ms_min": round(mn, 4), "ms_med": round(med, 4),
                          "ns_per_edge": round(mn * 1e6 / A.nnz, 4), "long_rows": A.plan.long_rows, "cut": A.plan.long_threshold,
                          "reordered": A.plan.reordered, "max_deg": int(deg.max())}), flush=True)
        if gen != "rmat":
            del A, D
            continue
        total = 0.0
        for lo, hi in ((1, 1), (2, 8), (9, 16), (17, 64), (65, 512), (513, 4096), (4097, 1 << 30)):
            rows = torch.nonzero((deg >= lo) & (deg <= hi)).flatten()
            if rows.numel() == 0:
                continue
            S = sub_csr(A, rows)
            S.plan
            Ds = torch.empty((S.n_rows, args.hidden), dtype=torch.float16, device=dev)
            mn, med = timed(lambda: ops.spmm(S, H, relu=True, out=Ds))
            total += mn
            print(json.dumps({"graph": gen, "class": f"deg {lo}..{hi if hi < (1 << 30) else 'max'}", "rows": S.n_rows, "edges": S.nnz,
                              "ms_min": round(mn, 4), "ms_med": round(med, 4), "ns_per_edge": round(mn * 1e6 / max(1, S.nnz), 4),
                              "share_of_edges": round(S.nnz / A.nnz, 4), "long_rows": S.plan.long_rows, "reordered": S.plan.reordered,
                              "GBps_alg": round((S.nnz * 134 + S.n_rows * 132) / mn / 1e6, 1)}), flush=True)
            del S, Ds
        print(json.dumps({"graph": gen, "class": "sum of the classes", "ms_min": round(total, 4)}), flush=True)


if __name__ == "__main__":
    main()
