#!/usr/bin/env python3
"""What building the row schedule (sgx_plan_create_ex, csrc/plan_build.hip) costs: wall time per call, stream
synchronised on both sides, for the S-100M adjacency (uniform and R-MAT), the arxiv shape and sampled mini-batch
sizes (the demo's NeighborLoader call pattern builds one per batch).

    python tools/plan_build_probe.py > gpurun_out/plan_build.jsonl
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgracex1_amd import graphs, ops  # noqa: E402


def timed(rowptr, reps, **kw):
    ops.Plan(rowptr, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan = ops.Plan(rowptr, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, plan


def main():
    cases = [("mini-batch 4 K nodes / 60 K edges (uniform)", lambda: graphs.uniform_graph(4096, 60_000), 200),
             ("mini-batch 40 K nodes / 1.2 M edges (rmat)", lambda: graphs.rmat_graph_n(40_000, 1_400_000), 100),
             ("arxiv shape 169 K / 2.3 M (rmat)", lambda: graphs.rmat_graph_n(169_343, 2_330_000), 100),
             ("S-100M 4.19 M / 104 M (uniform)", lambda: graphs.uniform_graph(1 << 22, 100_000_000), 20),
             ("S-100M 4.19 M / 104 M (rmat)", lambda: graphs.rmat_graph(22, 120_000_000), 20)]
    for name, make, reps in cases:
        A = make()
        ms, plan = timed(A.rowptr, reps)
        rec = {"graph": name, "rows": A.n_rows, "nnz": A.nnz, "plan_build_ms": round(ms, 4), "long_rows": plan.long_rows,
               "tasks": int(plan.export("task_row").numel()), "cut": plan.long_threshold, "reordered": plan.reordered,
               "natural_utilization": round(plan.natural_utilization, 4)}
        # what the round-1 builder did first: rowPtr to the host (its three passes over it on one core came on top)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            A.rowptr.cpu()
        rec["rowptr_to_host_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 4)
        print(json.dumps(rec), flush=True)
        del A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
