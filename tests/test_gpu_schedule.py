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
