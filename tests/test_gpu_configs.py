"""BASELINE.json's configurations 3-5 at their full sizes, on synthetic graphs of those shapes (the datasets are not
in the image): Reddit (233 K nodes, 114.6 M edges, dense 602 -> 128 -> 41), ogbn-products (2.45 M nodes, 123.7 M
edges, 100 -> 256 -> 47, both layer orders) and ogbn-arxiv GAT (169 K nodes, 2.3 M edges + self loops, 128 -> 256,
one head and 8 heads).  The two big ones are checked on sampled rows recomputed by the oracle from the rows of X
they touch; the GAT layer row by row and edge by edge against the oracle on the whole graph."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda")
HALF_BAND = dict(rtol=1e-2, atol=2e-3)          # fp16 storage / fp32 accumulation against the exact oracle


def _rand_w(p, m, gen):
    return ((torch.rand((p, m), generator=gen, device=dev) * 2 - 1) / p ** 0.5).half()


def _graph(gen_name, n, n_edges, seed):
    """uniform: row, col ~ U[0, n); rmat: R-MAT .57/.19/.19/.05 folded onto n nodes (the real Reddit /
    products / arxiv graphs are heavy-tailed -- hub rows, split tasks and the degree-ordered schedule at full size)."""
    from sgracex1_amd import graphs
    return (graphs.rmat_graph_n if gen_name == "rmat" else graphs.uniform_graph)(n, n_edges, seed=seed)


def _check_sampled_rows(oracle, A, X, Wt, D, relu, n_rows, gen, tol=HALF_BAND, n_longest=0):
    """Rows of D = act(A.(X.W)) recomputed from the rows of X behind their edges: H = X.W in fp32 on the host,
    rounded to half as the layer keeps it, summed by the oracle.  n_longest: the longest rows join the sample."""
    from _fixtures import sample_rows as _sample_rows
    rows = torch.randint(0, A.n_rows, (n_rows,), generator=gen, device=dev)
    if n_longest:
        rows = torch.cat([rows, torch.topk(A.rowptr.diff(), n_longest).indices])
    rows = rows.unique()
    srp, scol, sval, uniq = _sample_rows(A, rows)
    Hs = X[uniq].float().cpu().numpy() @ Wt.float().cpu().numpy().T
    Hs = Hs.astype(np.float16).astype(np.float32)
    want = oracle.spmm_f32(int(relu), (srp, scol, sval), Hs)
    np.testing.assert_allclose(D[rows].float().cpu().numpy(), want, **tol)
    return rows


@pytest.mark.parametrize("gen_name", ["uniform", "rmat"])
def test_config3_reddit_shape(oracle, gen_name):
    from sgracex1_amd import ops
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    n = 232_965
    A = _graph(gen_name, n, 114_600_000, 3)
    longest = 0
    if gen_name == "uniform":
        assert A.nnz > 114_000_000
    else:
        # coalescing removes the duplicates R-MAT draws on its hubs; what stays is heavy-tailed: hub rows cut into
        # tasks, the short rows scheduled in degree order
        assert A.nnz > 60_000_000 and A.plan.long_rows > 0 and int(A.rowptr.diff().max()) > 20_000
        longest = 64
    X = torch.rand((n, 602), generator=gen, device=dev).half()
    W1t, W2t = _rand_w(128, 602, gen), _rand_w(41, 128, gen)
    D1 = ops.layer_forward(A, X, W1t, relu=True)
    _check_sampled_rows(oracle, A, X, W1t, D1, True, 160, gen, n_longest=longest)
    D2 = ops.layer_forward(A, D1, W2t, relu=False)
    assert D2.shape == (n, 41)
    _check_sampled_rows(oracle, A, D1, W2t, D2, False, 160, gen, n_longest=longest)
    assert torch.isfinite(D2.float()).all() and torch.equal(D2, ops.layer_forward(A, D1, W2t, relu=False))
    del A, X, D1, D2
    torch.cuda.empty_cache()


@pytest.mark.parametrize("gen_name", ["uniform", "rmat"])
def test_config4_products_shape_both_orders(oracle, gen_name):
    from sgracex1_amd import ops
    gen = torch.Generator(device=dev)
    gen.manual_seed(4)
    n = 2_449_029
    A = _graph(gen_name, n, 123_700_000, 4)
    longest = 0
    if gen_name == "rmat":
        assert A.plan.long_rows > 0 and A.plan.reordered and int(A.rowptr.diff().max()) > 20_000
        longest = 64
    X = (torch.rand((n, 100), generator=gen, device=dev) - 0.2).half()
    W1t, W2t = _rand_w(256, 100, gen), _rand_w(47, 256, gen)
    D1 = ops.layer_forward(A, X, W1t, relu=True)
    rows = _check_sampled_rows(oracle, A, X, W1t, D1, True, 1024, gen, n_longest=longest)
    # the same layer aggregated first: inside the same band of the oracle, and close to the reference order
    S1 = ops.layer_forward(A, X, W1t, relu=True, order="aggregate_first")
    from _fixtures import sample_rows as _sample_rows
    srp, scol, sval, uniq = _sample_rows(A, rows)
    Hs = (X[uniq].float().cpu().numpy() @ W1t.float().cpu().numpy().T)
    want = oracle.spmm_f32(1, (srp, scol, sval), Hs)                      # unrounded H: the exact layer
    np.testing.assert_allclose(S1[rows].float().cpu().numpy(), want, **HALF_BAND)
    assert float((S1.float() - D1.float()).abs().max()) < (4e-3 if gen_name == "uniform" else 2e-2)
    assert torch.equal(S1, ops.layer_forward(A, X, W1t, relu=True, order="auto"))
    D2 = ops.layer_forward(A, D1, W2t, relu=False)
    assert D2.shape == (n, 47)
    _check_sampled_rows(oracle, A, D1, W2t, D2, False, 1024, gen, n_longest=longest)
    del A, X, D1, D2, S1
    torch.cuda.empty_cache()


@pytest.mark.parametrize("gen_name", ["uniform", "rmat"])
def test_config5_arxiv_shape_gat(oracle, gen_name):
    from sgracex1_amd import ops
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    n, P, heads = 169_343, 256, 8
    A = _graph(gen_name, n, 2_330_000, 5)                                 # + self loops: every row has a live edge
    if gen_name == "rmat":
        assert A.gat_plan.long_rows > 0                                   # hub rows: chunked softmax states, merged in order
    X = torch.rand((n, 128), generator=gen, device=dev).half()
    Wt = _rand_w(P, 128, gen)
    att = ((torch.rand(2 * P, generator=gen, device=dev) * 2 - 1) * 0.3).half()
    csr = (A.rowptr.cpu().numpy(), A.col.cpu().numpy(), A.val.float().cpu().numpy())
    Wh = ops.xw_dense(X, Wt)                                              # what the layer aggregates, rounded to half
    Wh_host = Wh.float().cpu().numpy()
    # one head, 256 wide: the reference's semantics
    got, E, S = ops.layer_forward(A, X, Wt, relu=True, gat_attention=att, want_edge_outputs=True)
    want, wE, wS = oracle.gat_f64(1, csr, Wh_host, att.float().cpu().numpy(), 0.2)
    np.testing.assert_allclose(got.float().cpu().numpy(), want, **HALF_BAND)
    np.testing.assert_allclose(E.cpu().numpy(), wE, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(S.cpu().numpy(), wS, rtol=4e-3, atol=1e-5)
    row = torch.repeat_interleave(torch.arange(n, device=dev), (A.rowptr[1:] - A.rowptr[:-1]).long())
    sums = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, row, S.double())
    assert (sums - 1).abs().max() < 1e-4                                  # a softmax per row
    # 8 heads of 32: the same formula per column slice with its own attention vector
    att8 = ((torch.rand((heads, 2 * (P // heads)), generator=gen, device=dev) * 2 - 1) * 0.3).half()
    got8 = ops.layer_forward(A, X, Wt, relu=True, gat_attention=att8.reshape(-1), gat_heads=heads)
    f = P // heads
    for h in (0, 3, 7):
        want_h, _e, _s = oracle.gat_f64(1, csr, np.ascontiguousarray(Wh_host[:, h * f:(h + 1) * f]),
                                        att8[h].float().cpu().numpy(), 0.2)
        np.testing.assert_allclose(got8[:, h * f:(h + 1) * f].float().cpu().numpy(), want_h, **HALF_BAND)
