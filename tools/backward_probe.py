#!/usr/bin/env python3
"""The products of the GCN layer's backward (FPYNQ.backward, MOL cell 16) at the Reddit and ogbn-arxiv shapes, fp32 as the
reference's backward: G = adj @ g (aggregation), grad_W = X^T @ G (sgx_xt_g), grad_x = G @ W^T (sgx_xw_dense), and the
ReLU mask -- what a training step on a large graph would spend beside the forward.

    python tools/backward_probe.py > gpurun_out/backward_probe.jsonl
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgracex1_amd import graphs, ops  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    for name, n, edges, M, P in (("ogbn-arxiv shape 169 K / 2.3 M, 128 -> 256", 169_343, 2_330_000, 128, 256),
                                 ("Reddit shape 233 K / 114.6 M, 602 -> 128", 232_965, 114_600_000, 602, 128)):
        A = graphs.uniform_graph(n, edges, seed=3, dtype=torch.float32)
        X = torch.rand((n, M), generator=g, device="cuda")
        Wt = torch.rand((P, M), generator=g, device="cuda") - 0.5
        grad = torch.rand((n, P), generator=g, device="cuda") - 0.5
        out = torch.rand((n, P), generator=g, device="cuda") - 0.3
        rec = {"shape": name, "nnz": A.nnz}
        rec["ms_relu_mask"] = round(timed(lambda: ops.relu_mask_backward(out, grad.clone())), 4) if hasattr(ops, "relu_mask_backward") else None
        G = ops.spmm(A, grad)
        rec["ms_adj_at_g_fp32"] = round(timed(lambda: ops.spmm(A, grad)), 4)
        rec["ms_xt_g"] = round(timed(lambda: ops.xt_g(X, G)), 4)
        rec["ms_g_wt"] = round(timed(lambda: ops.xw_dense(G, Wt.t().contiguous())), 4)
        rec["bytes_xt_g"] = n * (M + P) * 4
        rec["GBps_xt_g"] = round(rec["bytes_xt_g"] / rec["ms_xt_g"] / 1e6, 1)
        print(json.dumps(rec), flush=True)
        del A, X, G
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
