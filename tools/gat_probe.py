#!/usr/bin/env python3
"""The GAT aggregate (sgx_gat_aggregate: scores, softmax weights, weighted aggregation) on the shapes its stages are
tuned on, a few launches each -- for `rocprofv3 --kernel-trace --stats` (per-kernel times) or timed alone:

    python3 tools/gat_probe.py [arxiv|arxiv-rmat|rmat20] [--heads 1|8] [--launches 20]

SGX_GAT_SCAN=0 / 2 in the environment: stage A's short rows never / always in entry order (gat_scan.hip; default: by shape);
SGX_GAT_FUSED=0 / 2: the one-walk form (gat_fused.hip) never / wherever it applies (default: heads of up to 8 lanes).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgracex1_amd import graphs, ops  # noqa: E402
from sgracex1_amd.hipevents import Event  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="?", default="arxiv")
    ap.add_argument("--heads", type=int, default=1)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--width", type=int, default=0, help="columns of Wh (default: 256 on the arxiv shapes, 64 on rmat20)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    if a.shape == "arxiv":
        A, P = graphs.uniform_graph(169_343, 2_330_000, seed=5, device=dev), 256
    elif a.shape == "arxiv-rmat":
        A, P = graphs.rmat_graph_n(169_343, 2_330_000, seed=5, device=dev), 256
    else:
        A, P = graphs.rmat_graph(20, 30_000_000, seed=5, device=dev), 64
    P = a.width or P
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Wh = torch.rand((A.n_rows, P), generator=g, device=dev).half()
    att = ((torch.rand(2 * P, generator=g, device=dev) * 2 - 1) * 0.3).half()
    D = torch.empty((A.n_rows, P), dtype=torch.float16, device=dev)
    A.gat_plan, A.plan
    run = lambda: ops.gat_aggregate(A, Wh, att, relu=True, heads=a.heads, out=D)
    plain = lambda: ops.spmm(A, Wh, relu=True, out=D)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, fn in (("ms_gat_aggregate", run), ("ms_plain_aggregate", plain)):
        ts = []
        for _ in range(a.launches):
            b, e = Event(), Event()
            b.record(s)
            fn()
            e.record(s)
            ts.append(b.elapsed_ms(e))
        ts.sort()
        res[name] = round(ts[len(ts) // 2], 4)
    deg = A.rowptr.diff()
    print(json.dumps({"shape": a.shape, "heads": a.heads, "nodes": A.n_rows, "edges": A.nnz, "width": P, **res,
                      "max_degree": int(deg.max()), "rows_33_64": int(((deg > 32) & (deg <= 64)).sum()),
                      "rows_65_256": int(((deg > 64) & (deg <= 256)).sum()), "edges_65_256": int(deg[(deg > 64) & (deg <= 256)].sum()),
                      "rows_over_256": int((deg > 256).sum()), "edges_over_256": int(deg[deg > 256].sum()),
                      "gat_plan_long_rows": A.gat_plan.long_rows}), flush=True)


if __name__ == "__main__":
    main()
