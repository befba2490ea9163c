"""Backward on the device ("next" row f1): the weight gradient X^T @ G (sgx_xt_g), the transposed
CSR used when the layer's features are sparse, and a whole training step of the 2-layer model
against the torch twin."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("xdtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("n,M,P", [(5000, 64, 64), (100003, 64, 64), (4097, 7, 64), (3000, 100, 256), (2000, 1433, 16),
                                   (777, 65, 7), (40001, 602, 128), (9000, 128, 256), (5001, 36, 20)])
def test_xt_g_matches_fp64(xdtype, n, M, P):
    from sgracex1_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(n + M + P)
    X = torch.randn((n, M), generator=g, device="cuda").to(xdtype)
    G = torch.randn((n, P), generator=g, device="cuda")
    got = ops.xt_g(X, G)
    want = X.double().t() @ G.double()
    scale = X.double().abs().t() @ G.double().abs()
    assert got.shape == (M, P) and got.dtype == torch.float32
    assert ((got.double() - want).abs() / scale).max() < 5e-6
    again = ops.xt_g(X, G)
    assert torch.equal(got, again)                               # slab sums are order-fixed
    # the 16-byte-load kernel (aligned rows) and the scalar one it stands beside: every output sums its rows in the same order
    from sgracex1_amd import _lib
    with _lib.tuning(SGX_XTG_SCALAR="1"):
        scalar = ops.xt_g(X, G)
    assert torch.equal(got, scalar)


def test_csr_transpose_bit_exact():
    from sgracex1_amd import graphs, ops
    A = graphs.uniform_graph(3001, 40_000, seed=9, dtype=torch.float32, normalize=True)
    At = ops.csr_transpose(A)
    At.validate()
    dense = torch.zeros((A.n_rows, A.n_cols), device="cuda")
    row = torch.repeat_interleave(torch.arange(A.n_rows, device="cuda"), (A.rowptr[1:] - A.rowptr[:-1]).long())
    dense[row, A.col.long()] = A.val
    back = ops.Csr.from_dense(dense.t().contiguous(), torch.float32)
    assert torch.equal(At.rowptr, back.rowptr) and torch.equal(At.col, back.col) and torch.equal(At.val, back.val)


def test_training_step_gradients_match_torch_twin():
    """One optimisation step of GCN_PYNQ (sparse-X layer, ReLU, dense-X layer): every parameter
    gradient from the device backward against autograd through the dense fp32 formulation."""
    import os
    from _fixtures import GOLD
    from sgracex1_amd import molecule_gcn as M, pyg_lite as G, pynq_shim
    dev = torch.device("cuda")
    raw = np.load(os.path.join(GOLD, "mutag_raw.npz"))
    batch = G.collate(G.load_tu_raw(raw["A"], raw["graph_indicator"], raw["graph_labels"], raw["node_labels"])).to(dev)
    ip = pynq_shim.Overlay("gnn_all.bit").mmult_top_0
    model = M.GCN_PYNQ(64, 7, 2, ip).to(dev).eval()              # eval: no dropout, so both passes see the same net
    crit = torch.nn.CrossEntropyLoss()
    grads = {}
    for acc in (1, 0):
        model.zero_grad()
        crit(model(acc, batch.x, batch.edge_index, batch.batch), batch.y).backward()
        grads[acc] = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert set(grads[1]) == set(grads[0]) and "conv1.weight" in grads[1] and "conv2.weight" in grads[1]
    for k in grads[0]:
        ref = grads[0][k]
        err = (grads[1][k] - ref).abs().max() / (ref.abs().max() + 1e-12)
        assert err < 2e-2, (k, float(err))                        # fp16 forward vs fp32 twin


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_readout_mean_linear_matches_torch(dtype):
    """Fused global_mean_pool + Linear head ("next" row f3) against the two torch ops, with an empty
    graph and ragged graph sizes."""
    from sgracex1_amd import ops
    from sgracex1_amd.pyg_lite import global_mean_pool
    g = torch.Generator(device="cuda")
    g.manual_seed(2)
    sizes = torch.tensor([17, 1, 0, 300, 28, 5], device="cuda")
    batch = torch.repeat_interleave(torch.arange(6, device="cuda"), sizes)
    x = torch.randn((int(sizes.sum()), 64), generator=g, device="cuda").to(dtype)
    lin = torch.nn.Linear(64, 3).cuda()
    ptr = torch.zeros(7, dtype=torch.int32, device="cuda")
    ptr[1:] = torch.cumsum(sizes, 0)
    logits, pooled = ops.readout_mean_linear(x, ptr, lin.weight, lin.bias, want_pooled=True)
    want_pool = global_mean_pool(x.float(), batch, size=6)
    assert torch.allclose(pooled, want_pool, rtol=1e-5, atol=1e-6)
    assert torch.allclose(logits, lin(want_pool), rtol=1e-4, atol=1e-5)
    assert torch.equal(pooled[2], torch.zeros(64, device="cuda"))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("F", [64, 7, 300])
def test_readout_mean_pooling_in_the_training_step(dtype, F):
    """ops.ReadoutMean (the pooling of MOL cell 18 inside the training step: one launch each way) against
    global_mean_pool under autograd -- ragged graph sizes, an empty graph, the gradient in the layer output's own type;
    with rows that belong to no graph the gradient there is 0."""
    from sgracex1_amd import ops
    from sgracex1_amd.pyg_lite import global_mean_pool
    g = torch.Generator(device="cuda")
    g.manual_seed(F)
    sizes = torch.tensor([17, 1, 0, 300, 28, 5], device="cuda")
    batch = torch.repeat_interleave(torch.arange(6, device="cuda"), sizes)
    n = int(sizes.sum())
    x = torch.randn((n, F), generator=g, device="cuda").to(dtype).requires_grad_()
    twin = x.detach().clone().requires_grad_()
    up = torch.randn((6, F), generator=g, device="cuda")
    ptr = ops.graph_ptr_of(batch)
    assert ptr.dtype == torch.int32 and ptr.tolist() == [0, 17, 18, 18, 318, 346, 351] and ops.graph_ptr_of(batch) is ptr
    got = ops.ReadoutMean.apply(x, ptr, True)
    want = global_mean_pool(twin.float(), batch, size=6)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    got.backward(up)
    want.backward(up)
    assert x.grad.dtype == dtype
    torch.testing.assert_close(x.grad.float(), twin.grad.float(), rtol=1e-3 if dtype == torch.float16 else 1e-6, atol=1e-6)
    # a pointer that stops short of the last rows: their gradient is 0, not what the allocator left there
    short = ptr[:5].contiguous()
    junk = torch.full((n, F), float("nan"), device="cuda", dtype=dtype)
    del junk
    gx = ops.readout_mean_backward(up[:4], short, n, dtype)
    assert torch.equal(gx[:318], x.grad[:318]) and not gx[318:].any()


@pytest.mark.parametrize("F", [7, 16, 64, 256, 300])
@pytest.mark.parametrize("vdtype", [torch.float16, torch.float32])
def test_gat_backward_edge_pass_matches_dense_formulas(F, vdtype):
    """sgx_gat_backward_edges against the reference's dense backward (SG.py:884-1126 restated):
    softmax_out = g @ Wh^T, dx = S * softmax_out, sg = dx - S * rowsum(dx), mask, LeakyReLU slope."""
    from sgracex1_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(F)
    n, alpha = 700, 0.2
    adj = (torch.rand((n, n), generator=g, device=dev) < 0.02).float() * (torch.rand((n, n), generator=g, device=dev) + 0.1)
    adj[torch.arange(n), torch.arange(n)] = 1.0
    adj[torch.rand((n, n), generator=g, device=dev) < 0.002] = -0.5          # stored, masked out
    adj[3, :] = 0
    adj[3, torch.randperm(n, generator=g, device=dev)[:600]] = 0.3           # a long row
    # values a half can hold exactly, so that the fp16 forward sees the same Wh and a as the fp32 formulas
    Wh = torch.randn((n, F), generator=g, device=dev).half().float()
    att = (torch.randn((2 * F, 1), generator=g, device=dev) * (0.5 / F ** 0.5)).half().float()
    G = torch.randn((n, F), generator=g, device=dev)
    A = ops.Csr.from_dense(adj, vdtype)
    dense_adj = torch.zeros_like(adj)
    row = torch.repeat_interleave(torch.arange(n, device=dev), (A.rowptr[1:] - A.rowptr[:-1]).long())
    dense_adj[row, A.col.long()] = A.val.float()
    # forward quantities from the kernel itself (checked elsewhere), dense twins from torch
    _out, E, S = ops.gat_aggregate(A, Wh.to(vdtype), att.reshape(-1).to(vdtype), alpha=alpha, want_edge_outputs=True)
    e = torch.nn.functional.leaky_relu(Wh @ att[:F] + (Wh @ att[F:]).T, alpha)
    P = torch.softmax(torch.where(dense_adj > 0, e, torch.full_like(e, -9e15)), dim=1)
    dx = P * (G @ Wh.T)
    sg_d = dx - P * dx.sum(1, keepdim=True)
    sg_d = torch.where(dense_adj > 0, sg_d, torch.zeros_like(sg_d))
    sg_d = ((e > 0) + alpha * (e <= 0)) * sg_d
    sg, g1 = ops.gat_backward_edges(A, E, S, G, Wh, alpha)
    scale = float(sg_d.abs().max())
    assert torch.allclose(sg, sg_d[row, A.col.long()], rtol=2e-4, atol=2e-5 * scale)
    assert torch.allclose(g1, sg_d.sum(1), rtol=2e-4, atol=1e-4 * scale)
    # padded leading dimensions (what the layer's backward passes): same values
    Whp = torch.zeros((n, (F + 3) // 4 * 4 + 4), device=dev)
    Whp[:, :F] = Wh
    sg2, g12 = ops.gat_backward_edges(A, E, S, G, Whp[:, :F], alpha)
    assert torch.equal(sg2, sg) and torch.equal(g12, g1)


@pytest.mark.parametrize("F", [64, 256])
def test_gat_backward_edge_pass_power_law(F):
    """The edge pass on a heavy-tailed graph -- hub rows of thousands of entries (taken by whole wavefronts in the row
    kernel), rows without entries, an entry count that is not a multiple of the dots kernel's 256 -- against the same
    formulas on the edge list in float64."""
    from sgracex1_amd import graphs, ops
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(F)
    A = graphs.rmat_graph_n(30_011, 900_000, seed=F, self_loops=False)
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    assert int(deg.max()) > 2000 and int((deg == 0).sum()) > 0 and A.nnz % 256 != 0
    n, alpha = A.n_rows, 0.2
    val = A.val.float()
    val[torch.rand(A.nnz, generator=g, device=dev) < 0.01] = -0.5             # stored, masked out
    A = ops.Csr(A.rowptr, A.col, val.half(), A.n_cols)
    Wh = torch.randn((n, F), generator=g, device=dev).half().float()
    att = (torch.randn(2 * F, generator=g, device=dev) * (0.5 / F ** 0.5)).half()
    G = torch.randn((n, F), generator=g, device=dev)
    _out, E, S = ops.gat_aggregate(A, Wh.half(), att, alpha=alpha, want_edge_outputs=True)
    sg, g1 = ops.gat_backward_edges(A, E, S, G, Wh, alpha)
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    colj = A.col.long()
    d = (G.double()[row] * Wh.double()[colj]).sum(1)
    dx = S.double() * d
    rs = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, row, dx)
    want = dx - S.double() * rs[row]
    want = torch.where(A.val.float() > 0, want, torch.zeros_like(want))
    want = torch.where(E > 0, want, alpha * want)
    want_g1 = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, row, want)
    scale = float(want.abs().max())
    assert torch.allclose(sg.double(), want, rtol=2e-4, atol=2e-5 * scale)
    assert torch.allclose(g1.double(), want_g1, rtol=2e-4, atol=2e-4 * scale)
    assert not g1[deg == 0].any()
