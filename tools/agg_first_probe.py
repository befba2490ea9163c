#!/usr/bin/env python3
"""Layer 1 of the ogbn-products shape (F_in = 100 -> 256): the reference's order A . (X . W) against aggregating
the 100-wide input first, (A . X) . W -- 2.5x fewer gathered bytes; and the cost of gathering rows that start on
8 bytes (pitch 100 halves) against a 16-byte aligned copy (pitch 104).  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sgracex1_amd import graphs, ops  # noqa: E402
from tools.bench_configs import rand_w, timed  # noqa: E402

dev = torch.device("cuda")


def main():
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    n, f_in, hid = 2_449_029, 100, 256
    A = graphs.uniform_graph(n, 123_700_000, seed=4)
    A.plan
    X = torch.rand((n, f_in), generator=gen, device=dev).half()
    Xp = torch.zeros((n, 104), dtype=torch.float16, device=dev)
    Xp[:, :f_in] = X
    W1t = rand_w(hid, f_in, gen)
    H = ops.xw_dense(X, W1t)
    D = torch.empty((n, hid), dtype=torch.float16, device=dev)
    Z = torch.empty((n, f_in), dtype=torch.float16, device=dev)
    rec = {"config": "ogbn-products shape, layer 1", "nodes": n, "edges": A.nnz}
    rec["ms_xw_100_256"] = timed(lambda: ops.xw_dense(X, W1t), 10)
    rec["ms_agg_256"] = timed(lambda: ops.spmm(A, H, relu=True, out=D), 10)
    rec["ms_agg_100_pitch100"] = timed(lambda: ops.spmm(A, X, out=Z), 10)
    rec["ms_agg_100_pitch104"] = timed(lambda: ops.spmm(A, Xp, n_feat=f_in, out=Z), 10)
    ref = ops.spmm(A, H, relu=True)
    alt = torch.relu(ops.xw_dense(ops.spmm(A, X), W1t))
    rec["max_abs_diff_between_orders"] = float((ref.float() - alt.float()).abs().max())
    rec["max_abs_value"] = float(ref.float().abs().max())
    rec["ms_reference_order"] = rec["ms_xw_100_256"] + rec["ms_agg_256"]
    rec["ms_aggregate_first"] = rec["ms_agg_100_pitch100"] + rec["ms_xw_100_256"]
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
