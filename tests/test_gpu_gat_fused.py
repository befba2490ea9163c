"""The GAT aggregate as one walk over the rows (csrc/gat_fused.hip: the neighbours' scores formed from the rows it
gathers, a running softmax state per row and head, hardware exp) -- what runs when the caller wants no E / S.  **Parity
unpinned** like every GAT result.  Checked against the two-stage form of the same library (SGX_GAT_FUSED=0;
stated bound: 2e-3 relative in fp16 outputs -- their own rounding --, 1e-4 in fp32) and against the fp64 oracle of the
stored-edge formula: uniform and power-law graphs (degree order, tasks of long rows, rows without entries), 1 to 8 heads,
widths of 4 to 64 lanes per row, masked entries, rows without a live entry with and without the dense-emulation fill."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(gen_name, n, nnz, dtype, F, heads, seed, masked=0.03):
    from sgracex1_amd import graphs, ops
    A = (graphs.rmat_graph_n if gen_name == "rmat" else graphs.uniform_graph)(n, nnz, seed=seed, dtype=dtype, self_loops=(gen_name == "uniform"))
    g = torch.Generator(device="cuda")
    g.manual_seed(seed * 7 + F + heads)
    val = A.val.float()
    val[torch.rand(A.nnz, generator=g, device="cuda") < masked] = -0.25                 # stored, masked out of the softmax
    A = ops.Csr(A.rowptr, A.col, val.to(dtype), A.n_cols)
    A.plan
    Wh = (torch.randn((n, F), generator=g, device="cuda") * 0.6).to(dtype)
    att = (torch.randn(2 * F, generator=g, device="cuda") * (1.0 / (F // heads) ** 0.5)).to(dtype)
    return A, Wh, att


@pytest.mark.parametrize("gen_name", ["uniform", "rmat"])
@pytest.mark.parametrize("dtype,F,heads", [(torch.float16, 64, 1), (torch.float16, 64, 8), (torch.float16, 256, 8), (torch.float16, 256, 1),
                                           (torch.float16, 128, 4), (torch.float16, 32, 2), (torch.float16, 512, 1), (torch.float16, 48, 1),
                                           (torch.float32, 64, 1), (torch.float32, 256, 8), (torch.float32, 16, 2), (torch.float32, 100, 1)])
def test_fused_aggregate_matches_the_two_stages(gen_name, dtype, F, heads):
    from sgracex1_amd import _lib, ops
    A, Wh, att = _case(gen_name, 40_000, 1_300_000, dtype, F, heads, seed=3)
    if gen_name == "rmat":
        assert A.gat_plan.long_rows > 0 and A.gat_plan.reordered and int((A.rowptr[1:] == A.rowptr[:-1]).sum()) > 0
    tol = dict(rtol=2e-3, atol=1e-3) if dtype == torch.float16 else dict(rtol=1e-4, atol=1e-5)
    for fill in (False, True):
        junk = torch.full((8_000_000,), float("nan"), device="cuda")
        del junk
        with _lib.tuning(SGX_GAT_FUSED="2"):
            got = ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, fill_dead_rows=fill)
            again = ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, fill_dead_rows=fill)
        with _lib.tuning(SGX_GAT_FUSED="0"):
            ref = ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, fill_dead_rows=fill)
        assert torch.isfinite(got.float()).all() and torch.equal(got, again)
        torch.testing.assert_close(got.float(), ref.float(), **tol)
    if gen_name == "rmat" and F >= 64 and dtype == torch.float16:
        # the degree order's one-piece tail 64 rows per wavefront: the same fold per row, hence the same bits as without it
        with _lib.tuning(SGX_GAT_FUSED="2", SGX_SPMM_NO_SHORT_TAIL="1"):
            assert torch.equal(ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, fill_dead_rows=True), got)
    # rows without a live entry: exactly 0 without the fill
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(A.n_rows, device="cuda"), deg)
    live = torch.zeros(A.n_rows, dtype=torch.int64, device="cuda").index_add_(0, row, (A.val.float() > 0).long()) > 0
    with _lib.tuning(SGX_GAT_FUSED="2"):
        plain = ops.gat_aggregate(A, Wh, att, relu=False, heads=heads, fill_dead_rows=False)
    assert not plain[~live].any() and (gen_name == "uniform" or int((~live).sum()) > 0)


@pytest.mark.parametrize("dtype,F", [(torch.float16, 64), (torch.float32, 64), (torch.float16, 256)])
def test_fused_aggregate_against_the_oracle(oracle, dtype, F):
    from sgracex1_amd import _lib, ops
    A, Wh, att = _case("rmat", 30_000, 1_200_000, dtype, F, 1, seed=11)
    with _lib.tuning(SGX_GAT_FUSED="2"):
        got = ops.gat_aggregate(A, Wh, att, relu=True, fill_dead_rows=False)
    csr = (A.rowptr.cpu().numpy(), A.col.cpu().numpy(), A.val.float().cpu().numpy())
    want, _E, _S = oracle.gat_f64(1, csr, Wh.float().cpu().numpy(), att.float().cpu().numpy(), 0.2)
    tol = dict(rtol=1e-2, atol=2e-3) if dtype == torch.float16 else dict(rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(got.float().cpu().numpy(), want, **tol)


def test_fused_aggregate_small_graph_with_tasks():
    """a small matrix is cut at 64 entries: most of its rows' entries go through tasks and the finalize"""
    from sgracex1_amd import _lib, ops
    rng = np.random.default_rng(0)
    deg = np.concatenate([rng.integers(0, 9, 300), [65, 64, 1000, 5000, 0, 129], rng.integers(0, 200, 40)]).astype(np.int64)
    n = len(deg)
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    rowptr[1:] = torch.cumsum(torch.as_tensor(deg, device="cuda"), 0).int()
    nnz = int(rowptr[-1])
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    col = torch.randint(0, n, (nnz,), generator=g, device="cuda", dtype=torch.int32)
    val = (torch.rand(nnz, generator=g, device="cuda") - 0.1).half()
    A = ops.Csr(rowptr, col, val, n)
    A.plan
    assert A.gat_plan.long_rows > 0
    Wh = torch.randn((n, 64), generator=g, device="cuda").half()
    att = (torch.randn(128, generator=g, device="cuda") * 0.2).half()
    got = ops.gat_aggregate(A, Wh, att, relu=False, fill_dead_rows=False)              # (one head of 64 columns: the shape rule's own choice)
    with _lib.tuning(SGX_GAT_FUSED="0"):
        ref = ops.gat_aggregate(A, Wh, att, relu=False, fill_dead_rows=False)
    assert not torch.equal(got, ref)                                                    # (two forms, two roundings)
    torch.testing.assert_close(got.float(), ref.float(), rtol=2e-3, atol=1e-3)
