#!/usr/bin/env python3
"""Times the five BASELINE.json configurations (synthetic graphs of the stated shapes where the
dataset is not in the container) on one GPU and prints one JSON line per configuration:
per-stage times from hipEvents, edges/s, algorithmic GB/s of the aggregation stage.

    python tools/bench_configs.py [--only c1,c3] [--iters 20] [--gen uniform|rmat|both] [--pmc-launches K]

--gen: the generator of the c3 / c4 / c5 graphs (uniform: row, col ~ U[0, n); rmat: R-MAT .57/.19/.19/.05 folded onto
the shape's node count -- the real graphs are heavy-tailed).  --pmc-launches K: only K launches of each configuration's
aggregation kernel and nothing else, for a `rocprofv3 --pmc` pass (tools/pmc_passes.sh; summary: tools/configs_pmc.py).

c1  molecule_gcn: the 188-graph MUTAG batch (3371 nodes, 7442 edges), 7 -> 64 -> 64, sparse X then dense X
c2  Cora: the reference's cora_{adj,feat,weights}.txt, 1433 -> 64 (sparse X, ReLU) -> 7 (dense X)
c3  Reddit shape: N 232,965, 114.6 M edges (uniform synthetic), dense X 602 -> 128 (ReLU) -> 41
c4  ogbn-products shape: N 2,449,029, 123.7 M edges, dense X 100 -> 256 (ReLU) -> 47   (one GPU's view)
c5  ogbn-arxiv shape GAT: N 169,343, 2.33 M edges + self loops, 128 -> 8 x 32 = 256 wide single softmax
    (the reference's `nheads` only widens W, SG.py:1176-1178), relu = 1
Small configurations (c1, c2) are launch-bound, so they are also timed replayed from a hipGraph.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sgracex1_amd import graphs, ops  # noqa: E402
from sgracex1_amd.hipevents import Event  # noqa: E402

dev = torch.device("cuda")


def timed(fn, iters):
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


def graphed(fn, iters):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return timed(g.replay, iters)


def rand_w(p, m, gen):
    return ((torch.rand((p, m), generator=gen, device=dev) * 2 - 1) / p ** 0.5).half()


def report(name, A, stages, extra):
    rec = {"config": name, "nodes": A.n_rows, "edges": A.nnz}
    rec.update(stages)
    rec.update(extra)
    print(json.dumps(rec), flush=True)


PMC_LAUNCHES = 0


def make_graph(gen_name, n, n_edges, seed, **kw):
    return (graphs.rmat_graph_n if gen_name == "rmat" else graphs.uniform_graph)(n, n_edges, seed=seed, **kw)


def gcn_two_layer(name, A, X, W1t, W2t, iters, small=False):
    hid, out = W1t.shape[0], W2t.shape[0]
    D1 = torch.empty((A.n_rows, hid), dtype=torch.float16, device=dev)
    D2 = torch.empty((A.n_rows, out), dtype=torch.float16, device=dev)
    A.plan
    if isinstance(X, ops.Csr):
        X.plan
    if PMC_LAUNCHES:
        Hp = torch.empty((A.n_cols, hid), dtype=torch.float16, device=dev).normal_()
        for _ in range(PMC_LAUNCHES):
            ops.spmm(A, Hp, relu=True, out=D1)
        torch.cuda.synchronize()
        deg = A.rowptr.diff()
        print(json.dumps({"config": name, "pmc_launches": PMC_LAUNCHES, "nodes": A.n_rows, "edges": A.nnz, "hidden": hid,
                          "max_degree": int(deg.max()), "long_rows": A.plan.long_rows, "reordered": A.plan.reordered}), flush=True)
        return

    def fwd():
        ops.layer_forward(A, X, W1t, relu=True, out=D1)
        ops.layer_forward(A, D1, W2t, relu=False, out=D2)

    t_fwd = timed(fwd, iters)
    H1 = torch.empty((A.n_cols, hid), dtype=torch.float16, device=dev).normal_()
    H2 = torch.empty((A.n_cols, ops.table_pitch(out, 2)), dtype=torch.float16, device=dev).normal_()
    t_agg1 = timed(lambda: ops.spmm(A, H1, relu=True, out=D1), iters)
    t_agg2 = timed(lambda: ops.spmm(A, H2, relu=False, n_feat=out, out=D2), iters)
    if isinstance(X, ops.Csr):
        Wrm = ops.transpose(W1t)
        t_xw1 = timed(lambda: ops.spmm(X, Wrm, relu=False, out=D1), iters)
    else:
        t_xw1 = timed(lambda: ops.xw_dense(X, W1t), iters)
    t_xw2 = timed(lambda: ops.xw_dense(D1, W2t), iters)
    b_alg1 = A.nnz * (6 + hid * 2) + (A.n_rows + 1) * 4 + A.n_rows * hid * 2
    extra = {"f_in": W1t.shape[1], "hidden": hid, "out": out,
             "layer1": "sparse X" if isinstance(X, ops.Csr) else "dense X",
             "edges_per_s_2layer": 2 * A.nnz / (t_fwd * 1e-3),
             "agg1_algorithmic_GBps": b_alg1 / (t_agg1 * 1e-3) / 1e9,
             "agg1_frac_of_8TBps": b_alg1 / (t_agg1 * 1e-3) / 8e12,
             "agg1_algorithmic_bytes": b_alg1, "max_degree": int(A.rowptr.diff().max()),
             "plan": {"long_rows": A.plan.long_rows, "reordered": A.plan.reordered,
                      "natural_utilization": round(A.plan.natural_utilization, 3)}}
    stages = {"ms_forward_2layer": t_fwd, "ms_xw1": t_xw1, "ms_agg1": t_agg1, "ms_xw2": t_xw2, "ms_agg2": t_agg2}
    if not isinstance(X, ops.Csr) and W1t.shape[1] < hid:
        # input narrower than the hidden width: layer 1 aggregated first, D1 = relu((A.X).W1)  (order="auto")
        def fwd_auto():
            ops.layer_forward(A, X, W1t, relu=True, out=D1, order="auto")
            ops.layer_forward(A, D1, W2t, relu=False, out=D2, order="auto")
        stages["ms_forward_2layer_aggregate_first"] = timed(fwd_auto, iters)
        extra["edges_per_s_2layer_aggregate_first"] = 2 * A.nnz / (stages["ms_forward_2layer_aggregate_first"] * 1e-3)
    if small:
        stages["ms_forward_2layer_hipgraph"] = graphed(fwd, iters)
    report(name, A, stages, extra)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c1,c2,c3,c4,c5")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gen", default="uniform", choices=("uniform", "rmat", "both"))
    ap.add_argument("--pmc-launches", type=int, default=0)
    args = ap.parse_args()
    global PMC_LAUNCHES
    PMC_LAUNCHES = args.pmc_launches
    gens = ("uniform", "rmat") if args.gen == "both" else (args.gen,)
    want = set(args.only.split(","))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)

    if "c1" in want:
        from _fixtures import GOLD
        from sgracex1_amd import pyg_lite as G
        raw = np.load(os.path.join(GOLD, "mutag_raw.npz"))
        b = G.collate(G.load_tu_raw(raw["A"], raw["graph_indicator"], raw["graph_labels"], raw["node_labels"])).to(dev)
        A = ops.csr_from_edge_index(b.edge_index, b.num_nodes)
        X = ops.Csr.from_dense(b.x, torch.float16)
        gcn_two_layer("c1 molecule_gcn MUTAG batch", A, X, rand_w(64, 7, gen), rand_w(64, 64, gen), 200, small=True)
        # one training step of the notebook's model (forward on the kernels, backward on the kernels, Adam)
        from sgracex1_amd import molecule_gcn as MG, pynq_shim
        model = MG.GCN_PYNQ(64, 7, 2, pynq_shim.Overlay("gnn_all.bit").mmult_top_0).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        crit = torch.nn.CrossEntropyLoss()

        def train_step():
            opt.zero_grad()
            crit(model(1, b.x, b.edge_index, b.batch), b.y).backward()
            opt.step()

        print(json.dumps({"config": "c1 molecule_gcn training step (188 graphs)", "ms_train_step": timed(train_step, 50),
                          "reference": "RFSoC forward 14.8 + 39.4 ms, ARM backward 0.19-0.28 s per layer (MOL cell 20 output)"}),
              flush=True)

    if "c2" in want:
        from _fixtures import GOLD, load
        d = load("cora")
        w2 = np.load(os.path.join(GOLD, "cora.npz"))["w2"].astype(np.float32)
        A = graphs.csr_from_numpy(*d["adj"], d["N"])
        X = graphs.csr_from_numpy(*d["fea"], d["M_fea"])
        gcn_two_layer("c2 Cora (reference matrices)", A, X, torch.as_tensor(d["Wt"], device=dev).half(),
                      torch.as_tensor(np.ascontiguousarray(w2.T), device=dev).half(), 200, small=True)

    for g_name in (gens if "c3" in want else ()):
        n = 232_965
        A = make_graph(g_name, n, 114_600_000, 3)
        X = torch.rand((n, 602), generator=gen, device=dev).half()
        gcn_two_layer(f"c3 Reddit shape ({g_name})", A, X, rand_w(128, 602, gen), rand_w(41, 128, gen), args.iters)
        del A, X
        torch.cuda.empty_cache()

    for g_name in (gens if "c4" in want else ()):
        n = 2_449_029
        A = make_graph(g_name, n, 123_700_000, 4)
        X = torch.rand((n, 100), generator=gen, device=dev).half()
        gcn_two_layer(f"c4 ogbn-products shape, 1 GPU ({g_name})", A, X, rand_w(256, 100, gen), rand_w(47, 256, gen), args.iters)
        del A, X
        torch.cuda.empty_cache()

    for g_name in (gens if "c5" in want else ()):
        n, P = 169_343, 256
        A = make_graph(g_name, n, 2_330_000, 5)
        X = torch.rand((n, 128), generator=gen, device=dev).half()
        Wt = rand_w(P, 128, gen)
        att = ((torch.rand(2 * P, generator=gen, device=dev) * 2 - 1) * 0.3).half()
        D = torch.empty((n, P), dtype=torch.float16, device=dev)
        A.plan
        if PMC_LAUNCHES:
            Wh = ops.xw_dense(X, Wt)
            att8 = ((torch.rand(2 * P, generator=gen, device=dev) * 2 - 1) * 0.3).half()
            for _ in range(PMC_LAUNCHES):
                ops.gat_aggregate(A, Wh, att, relu=True, out=D)
            for _ in range(PMC_LAUNCHES):
                ops.gat_aggregate(A, Wh, att8, relu=True, heads=8, out=D)
            for _ in range(PMC_LAUNCHES):
                ops.spmm(A, Wh, relu=True, out=D)
            torch.cuda.synchronize()
            print(json.dumps({"config": f"c5 ogbn-arxiv shape GAT ({g_name})", "pmc_launches": PMC_LAUNCHES, "nodes": n,
                              "edges": A.nnz, "width": P, "max_degree": int(A.rowptr.diff().max()),
                              "long_rows": A.gat_plan.long_rows}), flush=True)
            continue
        t_layer = timed(lambda: ops.layer_forward(A, X, Wt, relu=True, gat_attention=att, out=D), 100)
        Wh = ops.xw_dense(X, Wt)
        t_gat = timed(lambda: ops.gat_aggregate(A, Wh, att, relu=True), 100)
        t_gcn = timed(lambda: ops.spmm(A, Wh, relu=True, out=D), 100)
        att8 = ((torch.rand(2 * P, generator=gen, device=dev) * 2 - 1) * 0.3).half()      # 8 vectors of 2 x 32
        t_layer8 = timed(lambda: ops.layer_forward(A, X, Wt, relu=True, gat_attention=att8, gat_heads=8, out=D), 100)
        t_gat8 = timed(lambda: ops.gat_aggregate(A, Wh, att8, relu=True, heads=8), 100)
        # edge pass of the backward (fp32, as the reference's backward): sampled g . Wh^T + softmax backward
        _o, E_, S_ = ops.gat_aggregate(A, Wh, att, relu=True, want_edge_outputs=True)
        Wh32, G32 = Wh.float(), torch.randn((n, P), generator=gen, device=dev)
        t_bwd = timed(lambda: ops.gat_backward_edges(A, E_, S_, G32, Wh32), 50)
        b_alg = A.nnz * (6 + 8 + P * 2) + (n + 1) * 4 + n * P * 2
        report(f"c5 ogbn-arxiv shape GAT ({g_name})", A,
               {"ms_layer": t_layer, "ms_gat_aggregate": t_gat, "ms_gcn_aggregate_same_shape": t_gcn,
                "ms_layer_8_heads": t_layer8, "ms_gat_aggregate_8_heads": t_gat8, "ms_gat_backward_edge_pass_fp32": t_bwd},
               {"f_in": 128, "width": P, "heads": "ms_layer: 8 x 32 as one 256-wide single-softmax head (reference semantics, "
                "nheads only widens W); ms_layer_8_heads: 8 independent softmaxes of 32 columns",
                "edges_per_s_layer": A.nnz / (t_layer * 1e-3), "gat_algorithmic_GBps": b_alg / (t_gat * 1e-3) / 1e9,
                "gat_algorithmic_bytes": b_alg, "max_degree": int(A.rowptr.diff().max()),
                "gat_over_plain_aggregate": t_gat / t_gcn, "long_rows": A.gat_plan.long_rows})

    if "c5" in want and not PMC_LAUNCHES:
        # the SGRACE library's own setting: float32 buffers, 8-bit quantised arithmetic (the reference's board
        # configs ship with fake_quantization = hardware_quantize = 1, w_qbits = 8)
        from sgracex1_amd import quant
        n, P = 169_343, 256
        A32 = graphs.uniform_graph(n, 2_330_000, seed=5, dtype=torch.float32)
        X32 = torch.rand((n, 128), generator=gen, device=dev)
        Wt32 = (torch.rand((P, 128), generator=gen, device=dev) * 2 - 1) / P ** 0.5
        att32 = (torch.rand(2 * P, generator=gen, device=dev) * 2 - 1) * 0.3
        D32 = torch.empty((n, P), dtype=torch.float32, device=dev)
        A32.plan
        qc = quant.constants(8)
        rec = {}
        for name, kw in (("gcn", {}), ("gat", {"gat_attention": att32})):
            rec[f"ms_layer_fp32_{name}"] = timed(lambda: ops.layer_forward(A32, X32, Wt32, relu=True, out=D32, **kw), 50)
            rec[f"ms_layer_fp32_{name}_8bit_quantised"] = timed(
                lambda: ops.layer_forward(A32, X32, Wt32, relu=True, out=D32, quant=qc, **kw), 50)
            rec[f"ms_layer_fp32_{name}_8bit_int8_operands"] = timed(
                lambda: ops.layer_forward(A32, X32, Wt32, relu=True, out=D32, quant=qc, quant_int8=True, **kw), 50)
            # what config.hardware_quantize = 1 runs (SGX_QUANT_INT8_AUTO): the faster of the two forms by shape
            rec[f"ms_layer_fp32_{name}_8bit_hardware_quantize"] = timed(
                lambda: ops.layer_forward(A32, X32, Wt32, relu=True, out=D32, quant=qc, quant_int8="auto", **kw), 50)
        report("c5 in the SGRACE library's setting (float32 buffers, w_qbits 8)", A32, rec, {"f_in": 128, "width": P})
        # the same graph with Reddit's 602 input features (a dense-feature layer wider than 128 columns: the shape the
        # integer operands are chosen for), 602 -> 128
        X602 = torch.rand((n, 602), generator=gen, device=dev)
        Wt602 = (torch.rand((128, 602), generator=gen, device=dev) * 2 - 1) / 128 ** 0.5
        D128 = torch.empty((n, 128), dtype=torch.float32, device=dev)
        rec = {"ms_layer_fp32_gcn": timed(lambda: ops.layer_forward(A32, X602, Wt602, relu=True, out=D128), 30),
               "ms_layer_fp32_gcn_8bit_quantised": timed(lambda: ops.layer_forward(A32, X602, Wt602, relu=True, out=D128, quant=qc), 30),
               "ms_layer_fp32_gcn_8bit_int8_operands": timed(
                   lambda: ops.layer_forward(A32, X602, Wt602, relu=True, out=D128, quant=qc, quant_int8=True), 30),
               "ms_layer_fp32_gcn_8bit_hardware_quantize": timed(
                   lambda: ops.layer_forward(A32, X602, Wt602, relu=True, out=D128, quant=qc, quant_int8="auto"), 30)}
        report("arxiv-sized graph, 602 dense input features -> 128, float32 buffers, w_qbits 8", A32, rec, {"f_in": 602, "width": 128})
        del A32, X32, D32, X602, D128
        # the X.W stage of a quantised layer alone on the Reddit shape (232 965 x 602 -> 128): fp32 values on the 8-bit
        # grid through the fp32 MFMA kernel against integer codes through the int8 matrix cores, with and without the
        # pass that quantises X (a caller that keeps its features as codes skips it)
        n, m, p = 232_965, 602, 128
        Xr = torch.rand((n, m), generator=gen, device=dev)
        Wr = (torch.rand((p, m), generator=gen, device=dev) * 2 - 1) / p ** 0.5
        Xq, Wq = ops.fake_quantize(Xr, 0, 8, qc.f_s, qc.f_z), ops.fake_quantize(Wr, 1, 8, qc.w_s, qc.w_z)
        Xc, _ = ops.quantize_codes_i8(Xr, 0, 8, qc.f_s, qc.f_z)
        Wc, _ = ops.quantize_codes_i8(Wr, 1, 8, qc.w_s, qc.w_z)
        rec = {"config": "quantised X.W alone, Reddit shape 232965 x 602 -> 128, 8 bit",
               "ms_fp32_quantise_X": timed(lambda: ops.fake_quantize(Xr, 0, 8, qc.f_s, qc.f_z, out=Xq), 30),
               "ms_fp32_xw": timed(lambda: ops.xw_dense(Xq, Wq), 30),
               "ms_int8_codes_of_X": timed(lambda: ops.quantize_codes_i8(Xr, 0, 8, qc.f_s, qc.f_z), 30),
               "ms_int8_xw": timed(lambda: ops.xw_dense_i8(Xc, Wc, m, 8, qc.scale_fea, qc.internal_quantization), 30)}
        rec["x_bytes_fp32"], rec["x_bytes_int8"] = n * m * 4, Xc.numel()
        print(json.dumps(rec), flush=True)
        del Xr, Xq, Xc

    if "c5" in want and not PMC_LAUNCHES:
        # the same layer on a power-law graph (R-MAT, 2^18 nodes): hub rows take the split path of the plan
        n, P = 1 << 18, 256
        A = graphs.rmat_graph(18, 2_330_000, seed=6)
        Wh = torch.rand((n, P), generator=gen, device=dev).half()
        att = ((torch.rand(2 * P, generator=gen, device=dev) * 2 - 1) * 0.05).half()
        A.plan
        deg = (A.rowptr[1:] - A.rowptr[:-1])
        t_split = timed(lambda: ops.gat_aggregate(A, Wh, att, relu=True), 50)
        t_plain = timed(lambda: ops.gat_aggregate(A, Wh, att, relu=True, use_plan=False), 20)
        report("c5 power-law variant: GAT aggregate, R-MAT 2^18 nodes", A,
               {"ms_gat_aggregate": t_split, "ms_gat_aggregate_without_plan": t_plain},
               {"width": P, "max_degree": int(deg.max()), "long_rows": A.plan.long_rows})


if __name__ == "__main__":
    main()
