"""Row schedules (sgx_plan): the degree-ordered packing used on power-law graphs must not change a
single bit of the result, since every row is still summed by one lane group in edge order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _skewed_csr(rng, n_rows, n_cols):
    # R-MAT-like: most rows tiny or empty, a few hundred-edge rows sprinkled in, a few long ones
    deg = np.where(rng.random(n_rows) < 0.6, 0, rng.integers(1, 6, n_rows))
    heavy = rng.choice(n_rows, n_rows // 9, replace=False)
    deg[heavy] = rng.integers(60, 400, heavy.size)
    deg[rng.choice(n_rows, 5, replace=False)] = rng.integers(600, 3000, 5)
    rp = np.zeros(n_rows + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    ci = rng.integers(0, n_cols, rp[-1]).astype(np.int32)
    va = (rng.standard_normal(rp[-1]) * 0.1).astype(np.float32)
    return rp, ci, va


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("P", [64, 24, 7])
def test_degree_ordered_schedule_is_bit_identical(dtype, P):
    from sgracex1_amd import ops
    rng = np.random.default_rng(P)
    n_rows, n_cols = 5003, 4000
    rp, ci, va = _skewed_csr(rng, n_rows, n_cols)
    A = ops.Csr(torch.as_tensor(rp, device="cuda"), torch.as_tensor(ci, device="cuda"),
                torch.as_tensor(va, device="cuda").to(dtype), n_cols)
    thr = A.plan.long_threshold
    assert thr == 64 and A.plan.long_rows == int((np.diff(rp) > thr).sum())           # a small matrix: early split
    assert A.plan.natural_utilization < 0.7 and A.plan.reordered
    H = torch.randn((n_cols, P), device="cuda").to(dtype)
    ordered = ops.spmm(A, H, relu=True, use_plan=True)
    natural = ops.spmm(A, H, relu=True, use_plan=False)
    short = torch.as_tensor(np.diff(rp) <= thr, device="cuda")
    assert torch.equal(ordered[short], natural[short])
    assert torch.allclose(ordered.float(), natural.float(), rtol=2e-3, atol=2e-3)
    assert not ordered[torch.as_tensor(np.diff(rp) == 0, device="cuda")].any()


def test_csr_from_edge_index_equals_dense_route():
    """Graph prep on the device: edge list -> CSR must equal to_dense_adj(...)._to_sparse_csr()."""
    from sgracex1_amd import ops
    from sgracex1_amd.pyg_lite import to_dense_adj
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    n = 700
    ei = torch.randint(0, n, (2, 5000), generator=g, device="cuda")
    ei = torch.cat([ei, ei[:, :300]], dim=1)                       # duplicates add up, as to_dense_adj does
    a = ops.csr_from_edge_index(ei, n, dtype=torch.float32)
    b = ops.Csr.from_dense(to_dense_adj(ei, n)[0], torch.float32)
    assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col, b.col) and torch.equal(a.val, b.val)
    a.validate()


def test_uniform_graph_keeps_natural_order():
    from sgracex1_amd import graphs
    A = graphs.uniform_graph(1 << 14, 400_000, seed=1)
    assert A.plan.natural_utilization > 0.7 and not A.plan.reordered and A.plan.long_rows == 0


@pytest.mark.parametrize("dtype,width,pitch", [
    (torch.float16, 100, 100),     # ogbn-products' F_in: 200-byte rows, 8-byte aligned
    (torch.float16, 602, 602),     # Reddit's F_in: 1204-byte rows, 4-byte aligned, wider than one 64-slot pass
    (torch.float16, 47, 50),       # a view: 47 columns used of 100-byte rows
    (torch.float16, 6, 6),         # 12-byte rows: every 16-byte gather runs into the next row
    (torch.float32, 7, 7),         # MUTAG's F_in in fp32: 28-byte rows
    (torch.float32, 25, 25),
    (torch.float16, 41, 41),       # rows on odd halves: ops copies the table into padded rows first
    (torch.float16, 5, 5),         # ... unless it is tiny: one element per lane (no vector gathers)
])
def test_tables_with_dword_aligned_rows(oracle, dtype, width, pitch):
    """Aggregation over tables whose rows start on a dword but not on 16 bytes (the input feature matrix of
    the backward products, unpadded hidden widths): vector gathers at dword alignment, hub rows through the
    split path, the last row of the table ending inside a gather -- against the oracle's fp32 sums, and
    the same bits as a 16-byte-aligned copy of the same table."""
    from sgracex1_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(width * 7 + pitch)
    n = 3000
    deg = torch.randint(0, 12, (n,), generator=g, device="cuda")
    deg[5] = 9000                                            # split path (plan cuts at 64 below 2^20 edges)
    deg[n - 1] = 700
    row = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    col = torch.randint(0, n, (row.numel(),), generator=g, device="cuda")
    col[-1] = n - 1                                          # the table's last row is gathered
    key = torch.unique(row * n + col)
    row, col = torch.div(key, n, rounding_mode="floor"), key % n
    val = (torch.rand(key.numel(), generator=g, device="cuda") * 0.2 + 0.01).to(dtype)
    A = ops.Csr.from_coo(row.to(torch.int32), col.to(torch.int32), val, n, n)
    store = (torch.rand((n, pitch), generator=g, device="cuda") - 0.4).to(dtype)
    H = store[:, :width]
    assert H.stride(0) == pitch
    got = ops.spmm(A, H, relu=True)
    assert got.shape == (n, width)
    csr = (A.rowptr.cpu().numpy(), A.col.cpu().numpy(), A.val.float().cpu().numpy())
    want = oracle.spmm_f32(1, csr, H.float().cpu().numpy())
    tol = dict(rtol=4e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got.float().cpu().numpy(), want, **tol)
    # 16-byte aligned copy of the table (padded pitch): the same fp32 chains, the same bits
    per16 = 8 if dtype == torch.float16 else 4
    padded = torch.zeros((n, (width + per16 - 1) // per16 * per16 + per16), dtype=dtype, device="cuda")
    padded[:, :width] = H
    assert torch.equal(ops.spmm(A, padded, relu=True, n_feat=width), got)
    # without the plan the hub rows are one chain each instead of 64-edge tasks: another fp32 summation order
    np.testing.assert_allclose(ops.spmm(A, H, relu=True, use_plan=False).float().cpu().numpy(), want, **tol)


@pytest.mark.parametrize("dtype,M_fea,P", [(torch.float16, 100, 256), (torch.float16, 200, 64), (torch.float16, 100, 47),
                                           (torch.float16, 101, 128), (torch.float32, 7, 64), (torch.float32, 130, 41)])
def test_aggregate_first_order(dtype, M_fea, P):
    """order = aggregate_first: D = act((A.X).W).  Bit-equal to its two stages run one by one (A.X rounded to the
    storage type, then the dense product with the ReLU on the rounded result), inside the layer's tolerance of
    the reference order, and selected by order="auto" exactly when M_fea < P."""
    from sgracex1_amd import graphs, ops
    g = torch.Generator(device="cuda")
    g.manual_seed(M_fea * 1000 + P)
    n = 50_021 if M_fea in (100, 101) else 20_011   # 100 columns x 50 K rows: the layer first copies X onto whole lines
    A = graphs.uniform_graph(n, 300_000, seed=M_fea + P, dtype=dtype)
    X = (torch.rand((n, M_fea), generator=g, device="cuda") - 0.3).to(dtype)
    Wt = ((torch.rand((P, M_fea), generator=g, device="cuda") * 2 - 1) / M_fea ** 0.5).to(dtype)
    for relu in (False, True):
        swapped = ops.layer_forward(A, X, Wt, relu=relu, order="aggregate_first")
        Z = ops.spmm(A, X)
        H = ops.xw_dense(Z, Wt)
        staged = torch.where(H > 0, H, torch.zeros_like(H)) if relu else H
        assert torch.equal(swapped, staged)
        ref = ops.layer_forward(A, X, Wt, relu=relu)
        tol = dict(rtol=1e-2, atol=2e-3) if dtype == torch.float16 else dict(rtol=1e-4, atol=1e-5)
        assert torch.allclose(swapped.float(), ref.float(), **tol), float((swapped.float() - ref.float()).abs().max())
        auto = ops.layer_forward(A, X, Wt, relu=relu, order="auto")
        assert torch.equal(auto, swapped if M_fea < P else ref)
    # captured into a hipGraph and replayed with new feature values: the same bits as the eager call
    from sgracex1_amd.graphed import Graphed
    out = torch.empty((n, P), dtype=dtype, device="cuda")
    run = Graphed(lambda: ops.layer_forward(A, X, Wt, relu=True, order="aggregate_first", out=out))
    X.mul_(-0.5)
    out.zero_()
    run()
    torch.cuda.synchronize()
    assert torch.equal(out, ops.layer_forward(A, X, Wt, relu=True, order="aggregate_first"))
    with pytest.raises(ValueError):
        ops.layer_forward(A, X, Wt, order="aggregate_first", acc_mode=ops.SGX_ACC_REF_HALF)
    with pytest.raises(ValueError):
        ops.layer_forward(A, X, Wt, order="columns_first")


def test_large_table_with_straddling_rows_is_repitched():
    """ops.spmm copies a large table of 200-byte rows onto whole 128-byte lines before a dense enough aggregation
    (32 edges per table row and more); the sums are the same chains, so the bits equal those of the table as given."""
    import ctypes
    from sgracex1_amd import graphs, ops
    from sgracex1_amd._lib import check, lib
    n, P = 50_000, 100
    A = graphs.uniform_graph(n, 2_000_000, seed=11)
    assert A.nnz >= 32 * n
    g = torch.Generator(device="cuda")
    g.manual_seed(12)
    X = (torch.rand((n, P), generator=g, device="cuda") - 0.5).half()
    assert ops._gatherable(X, P, A.nnz).stride(0) == 128 and ops._gatherable(X, P, 8 * n) is X
    got = ops.spmm(A, X, relu=True)
    raw = torch.empty((n, P), dtype=torch.float16, device="cuda")
    plan = A.plan
    sbytes = lib.sgx_spmm_scratch_bytes(plan.handle, P)
    scratch = torch.empty(max(1, sbytes), dtype=torch.uint8, device="cuda")
    check(lib.sgx_spmm_csr(0, 0, 1, 1, n, n, P, A.rowptr.data_ptr(), A.col.data_ptr(), A.val.data_ptr(), X.data_ptr(), P,
                           raw.data_ptr(), P, plan.handle, scratch.data_ptr(), sbytes,
                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "sgx_spmm_csr")
    assert torch.equal(got, raw)
