"""Seeded sweep over sgx_layer_forward: random shapes (rectangular adjacencies, one-row / one-column
cases, empty matrices, P and M_fea that are not multiples of anything), densities, both feature modes,
ReLU, GCN and GAT aggregates, fp16 and fp32 -- each case against the exact-math oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _csr_from_dense(d):
    rows, cols = np.nonzero(d)
    rp = np.zeros(d.shape[0] + 1, np.int32)
    rp[1:] = np.cumsum(np.bincount(rows, minlength=d.shape[0]))
    return rp, cols.astype(np.int32), d[rows, cols].astype(np.float32)


def _case(rng):
    N = int(rng.choice([1, 2, 3, 17, 64, 65, 200, 513, 1000]))
    square = rng.random() < 0.6
    M_adj = N if square else int(rng.choice([1, 5, 33, 300, 700]))
    M_fea = int(rng.choice([1, 2, 7, 16, 31, 64, 100, 129, 260]))
    P = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 21, 24, 41, 64, 65, 96, 128, 130, 256]))
    dens_a = float(rng.choice([0.0, 0.002, 0.02, 0.2, 0.9]))
    dens_x = float(rng.choice([0.0, 0.01, 0.1, 0.6]))
    return dict(N=N, M_adj=M_adj, M_fea=M_fea, P=P, dens_a=dens_a, dens_x=dens_x, gemm_mode=int(rng.random() < 0.5),
                relu=int(rng.random() < 0.5), gat=int(square and rng.random() < 0.35))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("seed", range(40))
def test_layer_forward_random_case(oracle, dtype, seed):
    from sgracex1_amd import ops
    rng = np.random.default_rng(1000 + seed)
    c = _case(rng)
    N, M_adj, M_fea, P = c["N"], c["M_adj"], c["M_fea"], c["P"]
    half = dtype == torch.float16
    rnd = (lambda a: oracle.to_half(np.asarray(a, np.float32)).astype(np.float32)) if half else (lambda a: np.asarray(a, np.float32))
    adj = rnd((rng.random((N, M_adj)) < c["dens_a"]) * (rng.random((N, M_adj)) * (0.3 if not c["gat"] else 1.0) + 0.05))
    if c["gat"]:
        adj[np.arange(N), np.arange(N)] = 1.0                     # every row keeps a live edge (sym_norm2's self loops)
        adj[rng.random((N, M_adj)) < 0.01] = rnd(-0.25)           # stored, masked out
        adj[np.arange(N), np.arange(N)] = 1.0
    x = rnd((rng.random((M_adj, M_fea)) < max(c["dens_x"], 0.0)) * rng.standard_normal((M_adj, M_fea)))
    Wt = rnd(rng.standard_normal((P, M_fea)) * (0.5 / np.sqrt(M_fea)))
    a_csr, x_csr = _csr_from_dense(adj), _csr_from_dense(x)
    dev = torch.device("cuda")

    def up(csr, n_cols):
        return ops.Csr(torch.as_tensor(csr[0], device=dev), torch.as_tensor(csr[1], device=dev),
                       torch.as_tensor(csr[2], device=dev).to(dtype), n_cols)

    A = up(a_csr, M_adj)
    X = up(x_csr, M_fea) if c["gemm_mode"] == 0 else torch.as_tensor(x, device=dev).to(dtype)
    fea = x_csr if c["gemm_mode"] == 0 else x
    tol = dict(rtol=1e-2, atol=3e-3) if half else dict(rtol=2e-5, atol=2e-5)
    if not c["gat"]:
        want = oracle.layer_f64(c["gemm_mode"], c["relu"], a_csr, fea, Wt, N=N, M_adj=M_adj, h_round=2 if half else 1)
        got = ops.layer_forward(A, X, torch.as_tensor(Wt, device=dev).to(dtype), relu=c["relu"])
        assert got.shape == (N, P)
        np.testing.assert_allclose(got.float().cpu().numpy(), want, err_msg=str(c), **tol)
        if c["gemm_mode"] == 1:
            # the same layer aggregated first, D = act((A.X).W): another association of the same sums
            swapped = ops.layer_forward(A, X, torch.as_tensor(Wt, device=dev).to(dtype), relu=c["relu"], order="aggregate_first")
            np.testing.assert_allclose(swapped.float().cpu().numpy(), want, err_msg="aggregate_first " + str(c), **tol)
            assert not swapped[torch.as_tensor(np.diff(a_csr[0]) == 0, device=dev)].any()
        else:
            with pytest.raises(ValueError):
                ops.layer_forward(A, X, torch.as_tensor(Wt, device=dev).to(dtype), relu=c["relu"], order="aggregate_first")
    else:
        att = rnd(rng.standard_normal(2 * P) * (0.5 / np.sqrt(P)))
        _, H = oracle.layer_f64(c["gemm_mode"], 0, a_csr, fea, Wt, N=N, M_adj=M_adj, h_round=2 if half else 1, return_h=True)
        want, wE, wS = oracle.gat_f64(c["relu"], a_csr, H, att, 0.2)
        got, E, S = ops.layer_forward(A, X, torch.as_tensor(Wt, device=dev).to(dtype), relu=c["relu"],
                                      gat_attention=torch.as_tensor(att, device=dev).to(dtype), want_edge_outputs=True)
        np.testing.assert_allclose(got.float().cpu().numpy(), want, err_msg=str(c), **tol)
        np.testing.assert_allclose(S.cpu().numpy(), wS, rtol=2e-3 if half else 1e-4, atol=1e-5, err_msg=str(c))
    # rows of A without entries give exactly +0 in the GCN aggregate
    if not c["gat"]:
        empty = np.diff(a_csr[0]) == 0
        assert not got[torch.as_tensor(empty, device=dev)].any()


@pytest.mark.parametrize("seed", range(30))
def test_exact_mode_random_case(oracle, seed):
    """SGX_ACC_REF_HALF over random shapes, densities, SPMM_BLOCK and thread splits, both feature modes:
    the layer's bits against the oracle's model of the reference's half arithmetic."""
    from sgracex1_amd import ops
    rng = np.random.default_rng(5000 + seed)
    N = int(rng.choice([5, 64, 333, 1000, 1500, 2100]))
    M_fea = int(rng.choice([1, 3, 8, 31, 64, 130, 257]))
    P = int(rng.choice([1, 7, 8, 16, 24, 25, 32, 41, 64, 100, 128, 200, 256]))
    spmm_block = int(rng.choice([1, 2, 3, 4, 8]))
    fea_threads, adj_threads = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 4]))
    gemm_mode, relu = int(rng.random() < 0.5), int(rng.random() < 0.5)
    adj = ((rng.random((N, N)) < float(rng.choice([0.005, 0.03, 0.3]))) * (rng.random((N, N)) - 0.3)).astype(np.float16)
    x = ((rng.random((N, M_fea)) < float(rng.choice([0.05, 0.5, 1.0]))) * rng.standard_normal((N, M_fea))).astype(np.float16)
    Wt = (rng.standard_normal((P, M_fea)) * (0.7 / np.sqrt(M_fea))).astype(np.float16)
    a_csr, x_csr = _csr_from_dense(adj.astype(np.float32)), _csr_from_dense(x.astype(np.float32))
    fea = x_csr if gemm_mode == 0 else x
    kw = dict(spmm_block=spmm_block, fea_threads=fea_threads, adj_threads=adj_threads)
    want = oracle.layer_refhalf(gemm_mode, relu, a_csr, fea, Wt, N=N, M_adj=N, **kw)
    dev = torch.device("cuda")

    def up(csr, n_cols):
        return ops.Csr(torch.as_tensor(csr[0], device=dev), torch.as_tensor(csr[1], device=dev),
                       torch.as_tensor(csr[2], device=dev).half(), n_cols)

    X = up(x_csr, M_fea) if gemm_mode == 0 else torch.as_tensor(x, device=dev)
    got = ops.layer_forward(up(a_csr, N), X, torch.as_tensor(Wt, device=dev), relu=relu, acc_mode=ops.SGX_ACC_REF_HALF, **kw)
    case = dict(N=N, M_fea=M_fea, P=P, gemm_mode=gemm_mode, relu=relu, **kw)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16)), case
