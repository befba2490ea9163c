#!/usr/bin/env python3
"""The reference's CPU formulation of the layer, timed on the GPU box's host cores beside the
oracle port that bench.py reports as `cpu_baseline` (SURVEY 8d "CPU baseline beside it"):

    torch.sparse.mm(A_csr, X @ W)   fp32, torch.set_num_threads(1) and (all cores)   -- `torch.spmm`/`adj @ input @ W`
                                     of the reference (paper Listing 1.3, MOL cell 17)
    scipy.sparse.csr_matrix @ ndarray  single thread                                   -- MMN cells 51-53
    the oracle's C loops on a thread pool                                              -- bench.py's "port"

on a row sample of the bench workload (rows [0, frac*N) of A.H at hidden 64, the full H table),
median of `--runs` after warm-up.  One JSON line per path.  Baselines only.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def median_time(fn, runs, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frac", type=float, default=0.125)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--workload", default="s100m")
    args = ap.parse_args()
    import bench
    from sgracex1_amd import graphs, ops
    dev = torch.device("cuda")
    wl = bench.WORKLOADS[args.workload]
    A, _X, _W1t, _W2t = bench.make_inputs(torch, graphs, ops, wl, 0, 1, dev)
    n, hidden = A.n_rows, wl["hidden"]
    rows = int(n * args.frac)
    e = int(A.rowptr[rows])
    rp = A.rowptr[:rows + 1].cpu()
    ci = A.col[:e].cpu()
    va = A.val[:e].float().cpu()
    H = torch.rand((n, hidden), generator=torch.Generator().manual_seed(1))
    del A
    torch.cuda.empty_cache()
    cores = os.cpu_count()
    base = {"workload": args.workload, "rows": rows, "edges": e, "hidden": hidden, "table_rows": n, "host_cpus": cores,
            "runs": args.runs}

    A_t = torch.sparse_csr_tensor(rp.long(), ci.long(), va, size=(rows, n))
    for k in (1, cores):
        torch.set_num_threads(k)
        t = median_time(lambda: torch.sparse.mm(A_t, H), args.runs)
        print(json.dumps(dict(base, path="torch.sparse.mm(A_csr, H) fp32", threads=k, seconds=t, edges_per_s=e / t)), flush=True)

    import scipy.sparse as sp
    A_s = sp.csr_matrix((va.numpy(), ci.numpy(), rp.numpy()), shape=(rows, n))
    Hn = H.numpy()
    t = median_time(lambda: A_s @ Hn, max(3, args.runs // 2))
    print(json.dumps(dict(base, path="scipy.sparse.csr_matrix @ ndarray fp32", threads=1, seconds=t, edges_per_s=e / t)),
          flush=True)

    import bench                                  # its cpu_baseline leg owns the oracle port
    rpn, cin, van = rp.numpy(), ci.numpy(), va.numpy()
    for k in (1, min(cores, 32)):
        t = median_time(bench.cpu_port_aggregate(rpn, cin, van, Hn, k), max(3, args.runs // 2))
        print(json.dumps(dict(base, path="oracle port (plain C loops, bench.py cpu_baseline kernel)", threads=k, seconds=t,
                              edges_per_s=e / t)), flush=True)


if __name__ == "__main__":
    main()
