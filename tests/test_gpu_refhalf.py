"""SGX_ACC_REF_HALF: the device reproduces the reference's HALF-build arithmetic bit for bit --
checked against the oracle's model on the reference's own matrices for every SPMM_BLOCK, both
feature modes, with and without ReLU, and against the csim log itself."""
import numpy as np
import pytest
import torch

from _fixtures import assert_prints_csim_log, known_answers, load

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    return t if dtype is None else t.to(dtype)


def _csr(ops, csr, n_cols):
    rp, ci, va = csr
    return ops.Csr(_dev(rp.astype(np.int32)), _dev(ci.astype(np.int32)), _dev(va.astype(np.float32), torch.float16), n_cols)


@pytest.mark.parametrize("name", ["mol", "cora", "citeseer", "test", "test2"])
@pytest.mark.parametrize("spmm_block", [1, 2, 4, 8])
def test_layer_bit_exact_sparse_features(oracle, name, spmm_block):
    from sgracex1_amd import ops
    d = load(name)
    A, X = _csr(ops, d["adj"], d["N"]), _csr(ops, d["fea"], d["M_fea"])
    Wt = _dev(d["Wt"], torch.float16)
    for relu in (0, 1):
        want = oracle.layer_refhalf(0, relu, d["adj"], d["fea"], d["Wt"], spmm_block=spmm_block)
        got = ops.layer_forward(A, X, Wt, relu=relu, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=spmm_block)
        assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("spmm_block", [1, 3, 4])
def test_layer_bit_exact_dense_features(oracle, spmm_block):
    """gemm_mode 1: zeros take part in the lane rotation (K.cpp:849-863), so this is a different
    summation order from the sparse stream -- layer 2 of cora (dense 64 -> 7) and mol (7 -> 64)."""
    from sgracex1_amd import ops
    import os
    from _fixtures import GOLD
    d = load("cora")
    w2 = np.load(os.path.join(GOLD, "cora.npz"))["w2"].astype(np.float32)
    h1 = oracle.layer_refhalf(0, 1, d["adj"], d["fea"], d["Wt"], spmm_block=spmm_block)
    want = oracle.layer_refhalf(1, 0, d["adj"], h1, np.ascontiguousarray(w2.T), spmm_block=spmm_block)
    A = _csr(ops, d["adj"], d["N"])
    got = ops.layer_forward(A, _dev(h1), _dev(np.ascontiguousarray(w2.T), torch.float16), relu=0,
                            acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=spmm_block)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))
    m = load("mol")
    want = oracle.layer_refhalf(1, 1, m["adj"], m["fea_dense"], m["Wt"], spmm_block=spmm_block)
    got = ops.layer_forward(_csr(ops, m["adj"], m["N"]), _dev(m["fea_dense"], torch.float16), _dev(m["Wt"], torch.float16),
                            relu=1, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=spmm_block)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))


def test_csim_log_from_the_device():
    """The device in reference arithmetic (SPMM_BLOCK 4) prints the csim log: 40 of 42 values to the digit, exactly
    the two known entries one ulp away (tests/_fixtures.py CSIM_RESIDUAL) -- the same text as the CPU model."""
    from sgracex1_amd import ops
    d = load("citeseer")
    got = ops.layer_forward(_csr(ops, d["adj"], d["N"]), _csr(ops, d["fea"], d["M_fea"]), _dev(d["Wt"], torch.float16),
                            relu=0, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4).cpu().numpy()
    assert_prints_csim_log({(r, j): "%g" % float(got[r, j]) for r in (0, 31) for j in range(21)})


def test_stage_entry_points_in_reference_arithmetic(oracle):
    from sgracex1_amd import ops
    d = load("cora")
    A = _csr(ops, d["adj"], d["N"])
    rng = np.random.default_rng(0)
    H = oracle.to_half(rng.standard_normal((d["N"], 24)).astype(np.float32))
    got = ops.spmm(A, _dev(H), relu=True, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4)
    eye = np.eye(24, dtype=np.float32)
    want = oracle.layer_refhalf(1, 1, d["adj"], H, eye, spmm_block=4)       # X.I is exact: isolates the A.H stage
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))
    with pytest.raises(RuntimeError):                                          # fp32 has no reference-half form
        ops.spmm(A.to(torch.float32), _dev(H).float(), acc_mode=ops.SGX_ACC_REF_HALF)


@pytest.mark.parametrize("name", ["mol", "cora", "test"])
@pytest.mark.parametrize("fea_threads,adj_threads,spmm_block", [(2, 2, 4), (4, 1, 4), (1, 4, 2), (4, 4, 1), (2, 4, 3)])
def test_layer_bit_exact_with_thread_splits(oracle, name, fea_threads, adj_threads, spmm_block):
    """FEA_THREADS / ADJ_THREADS of the reference (MM.h:166-167): contiguous row blocks, the sblock
    grouping restarts at each block (K.cpp:3159-3164, :3517-3523) -- sparse and dense features."""
    from sgracex1_amd import ops
    d = load(name)
    A, X = _csr(ops, d["adj"], d["N"]), _csr(ops, d["fea"], d["M_fea"])
    Wt = _dev(d["Wt"], torch.float16)
    kw = dict(spmm_block=spmm_block, fea_threads=fea_threads, adj_threads=adj_threads)
    want = oracle.layer_refhalf(0, 1, d["adj"], d["fea"], d["Wt"], **kw)
    got = ops.layer_forward(A, X, Wt, relu=1, acc_mode=ops.SGX_ACC_REF_HALF, **kw)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))
    if name == "cora" and spmm_block > 1:          # 2708 rows: every split above re-phases sblocks, tens of thousands of outputs move
        one = oracle.layer_refhalf(0, 1, d["adj"], d["fea"], d["Wt"], spmm_block=spmm_block)
        assert (one.view(np.uint16) != want.view(np.uint16)).sum() > 1000
    # dense features: the layer's own output fed back through a square weight matrix
    h = want.astype(np.float16)
    P = h.shape[1]
    rng = np.random.default_rng(3)
    w2t = (rng.standard_normal((P, P)) * 0.3).astype(np.float16)
    want2 = oracle.layer_refhalf(1, 0, d["adj"], h, w2t, **kw)
    got2 = ops.layer_forward(A, _dev(h), _dev(w2t), relu=0, acc_mode=ops.SGX_ACC_REF_HALF, **kw)
    assert np.array_equal(got2.cpu().numpy().view(np.uint16), want2.view(np.uint16))


@pytest.mark.parametrize("M,P", [(300, 64), (130, 24), (64, 256), (9, 40), (129, 8), (1433, 16)])
@pytest.mark.parametrize("spmm_block,fea_threads", [(1, 1), (4, 1), (3, 2)])
def test_dense_stage_bit_exact_random(oracle, M, P, spmm_block, fea_threads):
    """Dense X (every position of a row takes part, zeros included) through the lane-group kernel: several
    blocks of k, widths from one lane to 32 lanes per row, odd M -- against the oracle's model."""
    from sgracex1_amd import ops
    rng = np.random.default_rng(M * 7 + P)
    n = 1100
    x = (rng.standard_normal((n, M)) * (rng.random((n, M)) < 0.7)).astype(np.float16)
    wt = (rng.standard_normal((P, M)) * (1.0 / np.sqrt(M))).astype(np.float16)
    eye = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.ones(n, np.float32))
    want, H = oracle.layer_refhalf(1, 0, eye, x, wt, spmm_block=spmm_block, fea_threads=fea_threads, return_h=True)
    got = ops.xw_dense(_dev(x), _dev(wt), acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=spmm_block)
    if fea_threads == 1:
        assert np.array_equal(got.cpu().numpy().view(np.uint16), H.view(np.uint16))
    A = _csr(ops, eye, n)
    full = ops.layer_forward(A, _dev(x), _dev(wt), relu=0, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=spmm_block,
                             fea_threads=fea_threads)
    assert np.array_equal(full.cpu().numpy().view(np.uint16), want.view(np.uint16))


def test_reference_build_limits_exact_mode(oracle):
    """The public build's maxima (MM.h:43-45: MAX_N = MAX_M = 6144, MAX_P = 128) in one call, sparse features,
    SPMM_BLOCK 4, 2 FEA / 4 ADJ threads: the reference's half arithmetic bit for bit, and the default arithmetic
    inside its band of the exact result."""
    import torch
    from sgracex1_amd import ops
    rng = np.random.default_rng(6144)
    N = M = 6144
    P = 128

    def csr(rows, cols, per_row, scale):
        deg = rng.integers(0, 2 * per_row + 1, rows)
        deg[7] = cols                                                      # one full row
        deg[11] = 0
        rp = np.zeros(rows + 1, np.int32)
        rp[1:] = np.cumsum(deg)
        ci = np.concatenate([np.sort(rng.choice(cols, d, replace=False)) for d in deg]).astype(np.int32)
        va = ((rng.random(rp[-1]) - 0.3) * scale).astype(np.float16)
        return rp, ci, va

    a_csr, x_csr = csr(N, N, 12, 0.3), csr(N, M, 20, 1.0)
    Wt = (rng.standard_normal((P, M)) * (0.7 / np.sqrt(40))).astype(np.float16)
    kw = dict(spmm_block=4, fea_threads=2, adj_threads=4)
    want = oracle.layer_refhalf(0, 1, a_csr, x_csr, Wt, N=N, M_adj=N, **kw)
    dev = torch.device("cuda")

    def up(c, n_cols):
        return ops.Csr(torch.as_tensor(c[0], device=dev), torch.as_tensor(c[1], device=dev), torch.as_tensor(c[2], device=dev), n_cols)

    A, X, W = up(a_csr, N), up(x_csr, M), torch.as_tensor(Wt, device=dev)
    got = ops.layer_forward(A, X, W, relu=True, acc_mode=ops.SGX_ACC_REF_HALF, **kw)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))
    exact = oracle.layer_f64(0, 1, (a_csr[0], a_csr[1], a_csr[2].astype(np.float32)),
                             (x_csr[0], x_csr[1], x_csr[2].astype(np.float32)), Wt.astype(np.float32), N=N, M_adj=N, h_round=2)
    fast = ops.layer_forward(A, X, W, relu=True)
    np.testing.assert_allclose(fast.float().cpu().numpy(), exact, rtol=1e-2, atol=3e-3)
