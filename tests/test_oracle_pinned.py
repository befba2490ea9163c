"""Pins the CPU oracle to every known answer the reference holds for this path.

Sources (all relative to the reference checkout; transcribed into
tests/golden/known_answers.json by tests/golden/make_goldens.py):
  * csim log rows 0 / 31, citeseer, HALF build   (.../csim/report/mmult_top_csim.log:21-62)
  * hardware row 0 (fp16, P=16) and scipy row 0   (jupyter/test/mmult-master.ipynb cells 37, 55)
  * the 4x4 hand-checkable case                    (data/matrices/test_*.txt)
The reference C++ itself cannot be built in this image (Xilinx HLS headers absent).
"""
import numpy as np
import pytest

from _fixtures import assert_prints_csim_log, csr_to_dense, half_ulp_distance, known_answers, load


def _g(x):
    return "%g" % float(x)           # how std::cout prints a half (6 significant digits)


def test_kat_4x4(oracle):
    d = load("test")
    want = np.array(known_answers()["test_kat"]["D"], dtype=np.float32)
    got = oracle.layer_f64(0, 0, d["adj"], d["fea"], d["Wt"])
    assert np.array_equal(got, want)
    got_h = oracle.layer_refhalf(0, 0, d["adj"], d["fea"], d["Wt"])
    assert np.array_equal(got_h.astype(np.float32), want)
    # second hand case: A row0 = ones, X = ones(4x4), W = [[1,.5],[1,-.5],[1,.5],[1,-.5]]
    d2 = load("test2")
    got2 = oracle.layer_f64(0, 0, d2["adj"], d2["fea"], d2["Wt"])
    assert np.array_equal(got2, np.array([[16, 0], [0, 0], [0, 0], [0, 0]], dtype=np.float32))


def test_csim_log_citeseer(oracle):
    """HALF build, SPMM_BLOCK=4, FADD latency 4: 40 of the 42 logged values are reproduced to the printed digit; the
    other two -- (row 0, col 10) and (row 31, col 18), exactly these, one binary16 ulp each, the model's magnitude above
    the log's -- are printed by no setting or reading of the checked-in source (tests/csim_residual.py enumerates
    them; profiles/r02_csim_residual.txt), so the log comes from a kernel revision that differs from the source there.
    Every other (SPMM_BLOCK, latency, thread) setting reproduces fewer."""
    d = load("citeseer")
    ka = known_answers()["csim_log"]
    Wt = oracle.to_half(d["Wt"])
    D = oracle.layer_refhalf(0, 0, d["adj"], d["fea"], Wt, spmm_block=4, lat_fea=4, lat_adj=4)
    assert all(len(ka[r]) == 21 for r in ("0", "31"))
    assert_prints_csim_log({(int(r), j): _g(D[int(r), j]) for r in ("0", "31") for j in range(21)})
    exact = 40
    # the default build setting (SPMM_BLOCK=1) is a different, observable summation order
    D1 = oracle.layer_refhalf(0, 0, d["adj"], d["fea"], Wt, spmm_block=1)
    exact1 = sum(_g(D1[int(r), j]) == t for r in ("0", "31") for j, t in enumerate(ka[r]))
    assert exact1 < exact


def test_hardware_row0(oracle):
    d = load("citeseer")
    hw = known_answers()["hw_row0_fp16_P16"]
    Wt = oracle.to_half(np.ascontiguousarray(d["Wt"][:16]))
    D = oracle.layer_refhalf(0, 0, d["adj"], d["fea"], Wt, spmm_block=4)
    shown = [np.format_float_positional(x, unique=True) for x in D[0]]
    # the board printed the same (row 0, col 10) the csim log has: the one entry of this row the source does not give
    assert [j for j, (a, b) in enumerate(zip(shown, hw)) if not (a == b.rstrip("0") or a == b)] == [10]
    want = np.array([float(t) for t in hw], dtype=np.float16)
    assert half_ulp_distance(D[0], want).max() <= 1
    assert float(want[10]) == float(np.float16(float(known_answers()["csim_log"]["0"][10])))


def test_scipy_row0(oracle):
    """mmult-master.ipynb cell 55: csr(adj) @ (csr(fea) @ w) with every input parsed as float16."""
    d = load("citeseer")
    h = lambda a: oracle.to_half(a).astype(np.float32)
    adj = (d["adj_rowptr"], d["adj_col"], h(d["adj_val"]))
    fea = (d["fea_rowptr"], d["fea_col"], h(d["fea_val"]))
    D = oracle.layer_f64(0, 0, adj, fea, h(d["Wt"]))
    want = np.array([float(t) for t in known_answers()["scipy_row0_fp32_P21"]])
    np.testing.assert_allclose(D[0], want, rtol=2e-6, atol=1e-8)


@pytest.mark.parametrize("name", ["mol", "cora", "citeseer"])
def test_f64_matches_reference_software_check(oracle, name):
    d = load(name)
    got = oracle.layer_f64(0, 1, d["adj"], d["fea"], d["Wt"])
    want = oracle.layer_scipy(0, 1, d["adj"], d["fea"], d["w"])
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["mol", "cora", "citeseer"])
def test_refhalf_within_stated_band_of_exact(oracle, name):
    """the half kernel sits within atol=2e-3, rtol=1e-2 of exact math --
    the same band the device path is held to."""
    d = load(name)
    exact = oracle.layer_f64(0, 0, d["adj"], d["fea"], d["Wt"])
    for S in (1, 4):
        got = oracle.layer_refhalf(0, 0, d["adj"], d["fea"], d["Wt"], spmm_block=S)
        np.testing.assert_allclose(got.astype(np.float32), exact, rtol=1e-2, atol=2e-3)


def test_mol_sparse_and_dense_features_agree(oracle):
    """gemm_mode 0 (CSR X) and gemm_mode 1 (dense X) on the one-hot mol features."""
    d = load("mol")
    sp = oracle.layer_refhalf(0, 1, d["adj"], d["fea"], d["Wt"])
    de = oracle.layer_refhalf(1, 1, d["adj"], d["fea_dense"], d["Wt"])
    assert np.array_equal(sp.view(np.uint16), de.view(np.uint16))
    assert np.array_equal(csr_to_dense(d["fea"], (d["N"], d["M_fea"])), d["fea_dense"])
    e_sp = oracle.layer_f64(0, 1, d["adj"], d["fea"], d["Wt"])
    e_de = oracle.layer_f64(1, 1, d["adj"], d["fea_dense"], d["Wt"])
    np.testing.assert_allclose(e_sp, e_de, rtol=1e-6, atol=1e-7)


def test_relu_semantics(oracle):
    """K.cpp:2586-2590: keep v when v > 0 or relu == 0, otherwise write +0."""
    d = load("cora")
    a = oracle.layer_f64(0, 0, d["adj"], d["fea"], d["Wt"])
    b = oracle.layer_f64(0, 1, d["adj"], d["fea"], d["Wt"])
    assert (a < 0).any()
    assert np.array_equal(b, np.where(a > 0, a, np.float32(0)))
    bh = oracle.layer_refhalf(0, 1, d["adj"], d["fea"], d["Wt"], spmm_block=4)
    ah = oracle.layer_refhalf(0, 0, d["adj"], d["fea"], d["Wt"], spmm_block=4)
    assert np.array_equal(bh, np.where(ah > 0, ah, np.float16(0)))
    assert not np.signbit(bh).any()


def test_threads_split_matches_single_when_sblock_1(oracle):
    """FEA/ADJ_THREADS only re-partition rows (K.cpp:3159-3164, :3517-3523); with
    SPMM_BLOCK=1 the numbers cannot change."""
    d = load("cora")
    base = oracle.layer_refhalf(0, 1, d["adj"], d["fea"], d["Wt"])
    for ft, at in ((2, 2), (4, 4), (4, 2)):
        got = oracle.layer_refhalf(0, 1, d["adj"], d["fea"], d["Wt"], fea_threads=ft, adj_threads=at)
        assert np.array_equal(base.view(np.uint16), got.view(np.uint16))


def test_empty_rows_and_ragged_tail(oracle):
    rp = np.array([0, 0, 3, 3, 4, 4], dtype=np.int32)             # rows 0, 2, 4 empty; N=5 (not /4)
    ci = np.array([0, 2, 4, 1], dtype=np.int32)
    va = np.array([1.0, -2.0, 0.5, 3.0], dtype=np.float32)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((5, 3)).astype(np.float32)
    W = rng.standard_normal((3, 5)).astype(np.float32)
    Wt = np.ascontiguousarray(W.T)
    want = csr_to_dense((rp, ci, va), (5, 5)) @ (X @ W)
    got = oracle.layer_f64(1, 0, (rp, ci, va), X, Wt)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    assert not got[[0, 2, 4]].any()
    for S in (1, 2, 4, 8):
        gh = oracle.layer_refhalf(1, 0, (rp, ci, va), X, Wt, spmm_block=S)
        np.testing.assert_allclose(gh.astype(np.float32), want, rtol=2e-2, atol=2e-2)
        assert not gh[[0, 2, 4]].view(np.uint16).any()           # exactly +0


def test_gat_matches_dense_emulation(oracle):
    """SG.py:309-314, :634-661 restated densely (the way the reference's CPU path does it)."""
    rng = np.random.default_rng(1)
    N, F = 37, 8
    dense = (rng.random((N, N)) < 0.15).astype(np.float32) * rng.random((N, N)).astype(np.float32)
    dense[np.arange(N), np.arange(N)] = 1.0                         # self loops, as sym_norm2 adds
    rp = np.zeros(N + 1, dtype=np.int32)
    ci, va = [], []
    for i in range(N):
        nz = np.nonzero(dense[i])[0]
        ci += list(nz)
        va += list(dense[i, nz])
        rp[i + 1] = len(ci)
    adj = (rp, np.array(ci, np.int32), np.array(va, np.float32))
    Wh = rng.standard_normal((N, F)).astype(np.float32)
    att = rng.standard_normal((2 * F, 1)).astype(np.float32)
    alpha = 0.2
    e = Wh @ att[:F] + (Wh @ att[F:]).T
    e = np.where(e > 0, e, alpha * e)
    a1 = np.where(dense > 0, e, -9e15)
    a1 = a1 - a1.max(axis=1, keepdims=True)
    sm = np.exp(a1) / np.exp(a1).sum(axis=1, keepdims=True)
    want = sm @ Wh
    for relu in (0, 1):
        D, E, S = oracle.gat_f64(relu, adj, Wh, att, alpha)
        np.testing.assert_allclose(D, np.maximum(want, 0) if relu else want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(E, e[np.repeat(np.arange(N), np.diff(rp)), adj[1]], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(S, sm[np.repeat(np.arange(N), np.diff(rp)), adj[1]], rtol=1e-5, atol=1e-7)
