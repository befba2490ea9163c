#!/usr/bin/env python3
"""Where to cut long rows as a function of the graph's size: times the plain aggregate and the GAT aggregate (one head,
8 heads) on power-law graphs of several sizes for every cut (rows over `cut` edges -> tasks of `cut` edges)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sgracex1_amd import graphs, ops  # noqa: E402
from sgracex1_amd.hipevents import Event  # noqa: E402

dev = torch.device("cuda")


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    b, e = Event(), Event()
    b.record(s)
    for _ in range(iters):
        fn()
    e.record(s)
    return b.elapsed_ms(e) / iters


shapes = [("arxiv-rmat", 169_343, 2_330_000, 256), ("rmat 2^18 / 8 M", 1 << 18, 8_000_000, 128), ("rmat 2^20 / 30 M", 1 << 20, 30_000_000, 64)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0].startswith(sys.argv[1])]
g = torch.Generator(device=dev)
g.manual_seed(1)
for name, n, e, P in shapes:
    A = graphs.rmat_graph_n(n, e, seed=5)
    H = torch.rand((n, P), generator=g, device=dev).half()
    att = ((torch.rand(2 * P, generator=g, device=dev) * 2 - 1) * 0.3).half()
    D = torch.empty((n, P), dtype=torch.float16, device=dev)
    rec = {"graph": name, "nodes": n, "edges": A.nnz, "width": P, "max_degree": int(A.rowptr.diff().max())}
    for cut in [int(c) for c in os.environ.get("CUTS", "64,128,256,512,1024,2048,4096").split(",")]:
        plan = ops.Plan(A.rowptr, cut, cut)
        A._plan = plan
        A._gat_plan = plan
        rec[f"cut{cut}"] = {"long_rows": plan.long_rows,
                            "ms_plain": round(timed(lambda: ops.spmm(A, H, relu=True, out=D)), 4),
                            "ms_gat": round(timed(lambda: ops.gat_aggregate(A, H, att, relu=True, out=D)), 4),
                            "ms_gat8": round(timed(lambda: ops.gat_aggregate(A, H, att, relu=True, heads=8, out=D)), 4)}
    print(json.dumps(rec), flush=True)
