"""X.W for a large CSR X with the weight slice resident in LDS (csrc/xw_sparse_lds.hip; the reference's B_accel
tile, K.cpp:1960-2078, :3038-3051), through the C ABI (sgx_xw_sparse and sgx_layer_forward).

Checks: (i) identical bits to the gather kernel (the A.H entry point over the same operands forms the same fp32
fma chain per output element), (ii) sampled rows against the exact-math oracle, (iii) the shapes that select each
lane split / slice count, empty rows, rows longer than the entry ring, degree-ordered plans, W with inf in it.
"""
import numpy as np
import pytest
import torch

from _fixtures import sample_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sgx():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from sgracex1_amd import ops
    return ops


def _random_x(n, m, mean_deg, dtype, seed, empty_frac=0.05, long_every=0, long_deg=0, powerlaw=False):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    if powerlaw:
        deg = (torch.rand(n, generator=g, device="cuda") ** -0.9).clamp(max=min(m, 480)).long()   # under the plan's cut
    else:
        deg = torch.poisson(torch.full((n,), float(mean_deg), device="cuda"), generator=g).long()
    deg[torch.rand(n, generator=g, device="cuda") < empty_frac] = 0
    if long_every:
        deg[::long_every] = long_deg
    deg.clamp_(max=m)
    rp = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    rp[1:] = torch.cumsum(deg, 0)
    nnz = int(rp[-1])
    col = torch.randint(0, m, (nnz,), generator=g, device="cuda", dtype=torch.int32)
    val = (torch.rand(nnz, generator=g, device="cuda") * 2 - 1).to(dtype)
    return rp.to(torch.int32), col, val


# (dtype, M_fea, P, what the shape selects)
SHAPES = [
    (torch.float16, 1433, 64, "Cora: 4 lanes x 2 slices"),
    (torch.float16, 500, 64, "whole W resident: 8 lanes, 1 slice, two workgroups per CU"),
    (torch.float16, 3703, 21, "citeseer: 2 lanes x 2 slices, ragged last slice"),
    (torch.float16, 1433, 47, "ragged last slice, element-aligned stores"),
    (torch.float16, 700, 8, "one lane per row"),
    (torch.float16, 300, 200, "16 lanes x 2 slices"),
    (torch.float32, 1433, 64, "fp32: 4 lanes x 4 slices"),
    (torch.float32, 300, 16, "fp32: 4 lanes, 1 slice"),
    (torch.float32, 1000, 41, "fp32 ragged"),
]


@pytest.mark.parametrize("dtype,M,P,_what", SHAPES)
def test_lds_path_equals_gather_kernel_and_oracle(sgx, oracle, dtype, M, P, _what):
    n = 70_001
    rp, col, val = _random_x(n, M, 18, dtype, seed=M * 131 + P, long_every=997, long_deg=61)
    X = sgx.Csr(rp, col, val, M)
    assert X.nnz >= 1 << 20 and X.plan.long_rows == 0
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    W = ((torch.rand((M, P), generator=g, device="cuda") * 2 - 1) / 8).to(dtype)
    got = sgx.xw_sparse(X, W)
    ref = sgx.spmm(X, W, relu=False)                       # the gather kernel (A.H entry point), same operands
    assert torch.equal(got, ref)
    # sampled rows (the long ones and the empty ones among them) against exact math
    rows = torch.cat([torch.arange(0, n, 997, device="cuda")[:40], torch.randint(0, n, (400,), device="cuda")])
    srp, scol, sval, uniq = sample_rows(X, rows)
    table = W[uniq].float().cpu().numpy()
    eye = (np.arange(len(rows) + 1, dtype=np.int32), np.arange(len(rows), dtype=np.int32), np.ones(len(rows), np.float32))
    want = oracle.layer_f64(0, 0, eye, (srp, scol, sval), np.ascontiguousarray(table.T),
                            h_round=2 if dtype == torch.float16 else 1)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got[rows].float().cpu().numpy(), want, **tol)
    empty = (rp[1:] == rp[:-1])
    assert int(empty.sum()) > 0 and not got[empty].any()


def test_lds_path_inside_the_layer(sgx):
    """sgx_layer_forward with a large CSR X takes the LDS form for its first stage: equal to the stages run one by
    one through the gather kernel (H in the layer's pitch is rounded the same way)."""
    n, M, P = 80_000, 1433, 64
    rp, col, val = _random_x(n, M, 18, torch.float16, seed=5)
    X = sgx.Csr(rp, col, val, M)
    eye = sgx.Csr(torch.arange(n + 1, dtype=torch.int32, device="cuda"), torch.arange(n, dtype=torch.int32, device="cuda"),
                  torch.ones(n, dtype=torch.float16, device="cuda"), n)
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    Wt = ((torch.rand((P, M), generator=g, device="cuda") * 2 - 1) / 8).half()
    got = sgx.layer_forward(eye, X, Wt, relu=True)
    H = sgx.spmm(X, sgx.transpose(Wt), relu=False)
    assert torch.equal(got, torch.where(H > 0, H, torch.zeros_like(H)))


def test_lds_path_degree_ordered_plan(sgx):
    """A power-law X: the plan schedules rows in degree order (row_order); rows of hundreds of entries walk the
    entry ring many times."""
    n, M, P = 300_000, 1433, 64
    rp, col, val = _random_x(n, M, 0, torch.float16, seed=9, powerlaw=True)
    X = sgx.Csr(rp, col, val, M)
    assert X.nnz >= 1 << 20
    assert X.plan.long_rows == 0 and int(rp.diff().max()) > 400
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    W = ((torch.rand((M, P), generator=g, device="cuda") * 2 - 1) / 8).half()
    assert torch.equal(sgx.xw_sparse(X, W), sgx.spmm(X, W, relu=False))


def test_lds_path_empty_slots_read_the_zero_row(sgx):
    """Slots past the end of a row read the tile's zero row with weight 0: a non-finite W[0] (or any other row)
    must not leak into rows that do not reference it."""
    n, M, P = 70_000, 1433, 64
    rp, col, val = _random_x(n, M, 18, torch.float16, seed=21)
    col = col.clamp(min=1)                                  # nobody references column 0
    X = sgx.Csr(rp, col, val, M)
    W = torch.full((M, P), 0.25, dtype=torch.float16, device="cuda")
    W[0] = float("inf")
    got = sgx.xw_sparse(X, W)
    assert torch.isfinite(got.float()).all()
