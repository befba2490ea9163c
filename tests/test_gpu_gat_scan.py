"""Stage A of the two-stage GAT aggregate in entry order (csrc/gat_scan.hip: a segmented scan over row-aligned windows of the
stored entries for the rows of up to 256 entries; longer rows through the plan's tasks, their scores left in W for stage B)
-- the softmax weights of SG.py:634-657 on the stored entries.  **Parity unpinned** like every GAT number here (SURVEY
8c: the reference holds no fixture for its attention path).  What is checked: the weights against an fp64 softmax of the
kernel's own scores E (stated bound: 1e-5 relative -- the maximum is exact, the sum is added in scan order), the
row-shaped kernels of the same stage (SGX_GAT_SCAN=0 against =2: E bit for bit, weights and rows inside the same bound),
the same bits run to run and with / without the side outputs, and the structural cases a window can meet: rows that begin
exactly on a window, windows inside a hub, hubs as a window's last row, runs of 64 and more rows without entries, one-entry
rows (more than 64 row starts per window), masked entries, rows without a live entry, matrices of 1 / 63 / 64 / 65 entries."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _csr(deg, n_cols, dtype, seed, masked=0.05, dead_rows=()):
    from sgracex1_amd import ops
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    deg = torch.as_tensor(deg, dtype=torch.int64, device="cuda")
    rowptr = torch.zeros(deg.numel() + 1, dtype=torch.int64, device="cuda")
    rowptr[1:] = torch.cumsum(deg, 0)
    nnz = int(rowptr[-1])
    col = torch.randint(0, n_cols, (nnz,), generator=g, device="cuda", dtype=torch.int32)
    val = torch.rand(nnz, generator=g, device="cuda") * 0.9 + 0.1
    val[torch.rand(nnz, generator=g, device="cuda") < masked] = -0.5          # stored, masked out of the softmax
    for r in dead_rows:
        val[int(rowptr[r]):int(rowptr[r + 1])] = 0.0
    A = ops.Csr(rowptr.to(torch.int32), col, val.to(dtype), n_cols)
    A.plan                                                                    # (small matrices run without a plan unless one exists)
    return A


def _reference_weights(A, E, heads):
    """fp64 softmax over each row's live entries of the kernel's own scores"""
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(A.n_rows, device="cuda"), deg)
    live = (A.val.float() > 0)
    x = E.double().reshape(A.nnz, heads)
    xm = torch.where(live[:, None], x, torch.full_like(x, -float("inf")))
    m = torch.full((A.n_rows, heads), -float("inf"), dtype=torch.float64, device="cuda")
    m = m.scatter_reduce(0, row[:, None].expand(-1, heads), xm, reduce="amax")
    p = torch.where(live[:, None], torch.exp(x - m[row]), torch.zeros_like(x))
    l = torch.zeros((A.n_rows, heads), dtype=torch.float64, device="cuda").index_add_(0, row, p)
    return torch.where(l[row] > 0, p / l[row], torch.zeros_like(p)), row, live


def _check(A, heads, f_head, dtype, seed, fill=None):
    from sgracex1_amd import _lib, ops
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    F = heads * f_head
    Wh = (torch.randn((A.n_cols, F), generator=g, device="cuda") * 0.7).to(dtype)
    att = (torch.randn(2 * F, generator=g, device="cuda") * (1.5 / f_head ** 0.5)).to(dtype)
    kw = dict(alpha=0.2, relu=True, heads=heads, want_edge_outputs=True, fill_dead_rows=fill)
    junk = torch.full((4_000_000,), float("nan"), device="cuda")               # scratch handed out next is not zeros
    del junk
    with _lib.tuning(SGX_GAT_SCAN="2"):
        D, E, S = ops.gat_aggregate(A, Wh, att, **kw)
    with _lib.tuning(SGX_GAT_SCAN="0"):
        D0, E0, S0 = ops.gat_aggregate(A, Wh, att, **kw)
    assert torch.equal(E, E0)
    want, row, live = _reference_weights(A, E, heads)
    S2 = S.reshape(A.nnz, heads)
    assert torch.isfinite(S2).all()
    torch.testing.assert_close(S2.double(), want, rtol=1e-5, atol=1e-30)
    torch.testing.assert_close(S0.reshape(A.nnz, heads).double(), want, rtol=1e-5, atol=1e-30)
    assert not S2[~live].any()
    tol = dict(rtol=2e-3, atol=1e-3) if dtype == torch.float16 else dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(D.float(), D0.float(), **tol)
    # run to run: the same bits (rows across ranges are merged in a fixed order)
    with _lib.tuning(SGX_GAT_SCAN="2", SGX_GAT_FUSED="0"):
        D1, E1, S1 = ops.gat_aggregate(A, Wh, att, **kw)
        assert torch.equal(S, S1) and torch.equal(D, D1)
        # and without the side outputs (the weights in scratch instead of the caller's S)
        assert torch.equal(ops.gat_aggregate(A, Wh, att, alpha=0.2, relu=True, heads=heads, fill_dead_rows=fill), D)
    return D, S2, row


def _filled(head, rng, n=58_000):
    """the case's rows, then ordinary rows up to 2^20 entries and more: smaller matrices are cut at 64 entries whatever the
    caller's plan says (plan_build.hip), and the entry windows exist for the cut at 256 only"""
    return np.concatenate([head, rng.integers(0, 40, n)])


DEGREES = {
    "mixed": lambda rng: rng.integers(0, 41, 60_000),
    "hubs": lambda rng: _filled(np.concatenate([np.full(1000, 5), [100_000], np.full(500, 3),
                                                [513, 512, 511, 256, 255, 257, 64, 63, 65, 1, 0, 0, 2], [3000, 0, 7, 40_000],
                                                rng.integers(0, 20, 3000)]), rng),
    "empty_runs": lambda rng: _filled(np.concatenate([np.zeros(200), rng.integers(1, 9, 500), np.zeros(300), [700], np.zeros(64), [1],
                                                      np.zeros(129), rng.integers(0, 3, 4000), np.zeros(1000)]), rng),
    "one_entry_rows": lambda rng: np.ones(1_100_000),
    "on_the_windows": lambda rng: _filled(np.concatenate([np.full(10, 512), np.full(10, 256), np.full(3, 1024), [64, 64, 128, 192, 64],
                                                          np.full(40, 128), np.full(7, 256), [255, 1, 256, 256, 257, 255]]), rng),
    "two_entry_rows_then_hub": lambda rng: _filled(np.concatenate([np.full(5000, 2), [20_000], np.full(5000, 2)]), rng),
    "long_rows_back_to_back": lambda rng: _filled(np.concatenate([[300, 300, 257, 256, 1, 5000, 4000, 0, 0, 256, 300], np.zeros(70),
                                                                  [300, 2, 300]]), rng),
    "rows_of_256": lambda rng: _filled(np.array([0, 256, 0, 0, 256, 255, 1, 0]), rng),
    # small matrices (cut at 64: the row-shaped kernels whatever SGX_GAT_SCAN says; without a longer row the windows exist)
    "one_row": lambda rng: np.array([5000]),
    "one_row_between_empty": lambda rng: np.array([0, 5000, 0]),
    "one_entry": lambda rng: np.array([0, 1, 0]),
    "63": lambda rng: np.array([63]),
    "64": lambda rng: np.array([30, 34]),
    "65": lambda rng: np.array([64, 1]),
}
SCANNED = ("mixed", "hubs", "empty_runs", "one_entry_rows", "on_the_windows", "two_entry_rows_then_hub", "long_rows_back_to_back",
           "rows_of_256", "one_entry", "63", "64", "65")


@pytest.mark.parametrize("name", list(DEGREES))
@pytest.mark.parametrize("dtype,heads,f_head", [(torch.float16, 1, 64), (torch.float16, 8, 32), (torch.float32, 1, 32),
                                                (torch.float16, 4, 16), (torch.float32, 2, 8), (torch.float16, 3, 8),
                                                (torch.float32, 16, 8)])
def test_scan_weights(name, dtype, heads, f_head):
    rng = np.random.default_rng(len(name) * 7 + heads)
    deg = DEGREES[name](rng).astype(np.int64)
    n_cols = max(len(deg), 50)
    A = _csr(deg, n_cols, dtype, seed=heads * 31 + f_head)
    assert (A.gat_plan.export("scan_win").numel() > 0) == (name in SCANNED)        # (the scan is what SGX_GAT_SCAN=2 runs)
    _check(A, heads, f_head, dtype, seed=heads + f_head, fill=False)


@pytest.mark.parametrize("dtype,heads,f_head", [(torch.float16, 1, 64), (torch.float16, 8, 16), (torch.float32, 2, 16)])
def test_scan_rows_without_a_live_entry(dtype, heads, f_head):
    """Rows whose stored entries are all masked, and rows without entries: weight 0 on every entry, and with the
    dense-emulation rule (SG.py:638-641) the mean row of Wh and S = 1/N -- a short row, a row of exactly one range, a hub
    over many windows, rows of the scan's own length limit, the last row."""
    rng = np.random.default_rng(5)
    deg = np.concatenate([rng.integers(1, 30, 3000), [512, 9000, 700, 256, 200], rng.integers(0, 40, 60_000), [40]]).astype(np.int64)
    dead = [0, 17, 3000, 3001, 3002, 3003, 3004, len(deg) - 1]
    A = _csr(deg, len(deg), dtype, seed=9, dead_rows=dead)
    assert A.gat_plan.export("scan_win").numel() > 0
    D, S, row = _check(A, heads, f_head, dtype, seed=3, fill=False)
    dead_t = torch.tensor(dead, device="cuda")
    assert not D[dead_t].any()
    assert not S[torch.isin(row, dead_t)].any()
    Df, Sf, _ = _check_fill(A, heads, f_head, dtype)
    empty = (A.rowptr[1:] == A.rowptr[:-1]).nonzero().flatten()
    for rows in (dead_t, empty):
        assert torch.isfinite(Df[rows]).all() and Df[rows].float().abs().sum() > 0
    assert torch.equal(Sf[torch.isin(row, dead_t)], torch.full_like(Sf[torch.isin(row, dead_t)], 1.0 / A.n_cols))


def _check_fill(A, heads, f_head, dtype):
    from sgracex1_amd import _lib, ops
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    F = heads * f_head
    Wh = (torch.randn((A.n_cols, F), generator=g, device="cuda") * 0.7 + 0.3).to(dtype)
    att = (torch.randn(2 * F, generator=g, device="cuda") / f_head ** 0.5).to(dtype)
    with _lib.tuning(SGX_GAT_SCAN="2"):
        D, E, S = ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, want_edge_outputs=True, fill_dead_rows=True)
    with _lib.tuning(SGX_GAT_SCAN="0"):
        D0, E0, S0 = ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, want_edge_outputs=True, fill_dead_rows=True)
    assert torch.equal(E, E0)
    torch.testing.assert_close(S, S0, rtol=1e-5, atol=1e-30)
    torch.testing.assert_close(D.float(), D0.float(), **(dict(rtol=2e-3, atol=1e-3) if dtype == torch.float16 else dict(rtol=1e-5, atol=1e-6)))
    mean_row = torch.relu(Wh.float().mean(0))
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(A.n_rows, device="cuda"), deg)
    live = torch.zeros(A.n_rows, dtype=torch.int64, device="cuda").index_add_(0, row, (A.val.float() > 0).long()) > 0
    torch.testing.assert_close(D[~live].float(), mean_row.expand(int((~live).sum()), F), rtol=1e-2, atol=2e-3)
    return D, S.reshape(A.nnz, heads), row


def test_scan_on_a_power_law_graph_against_the_oracle(oracle):
    """R-MAT (hubs of tens of thousands of entries, thousands of empty rows), one head: rows, E and S against the fp64
    oracle of the stored-edge formula."""
    from sgracex1_amd import graphs, ops
    n, F = 60_000, 64
    A = graphs.rmat_graph_n(n, 2_000_000, seed=4, dtype=torch.float16, self_loops=False)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    Wh = (torch.randn((n, F), generator=g, device="cuda") * 0.5).half()
    att = (torch.randn(2 * F, generator=g, device="cuda") * (0.5 / F ** 0.5)).half()
    assert A.gat_plan.reordered                                  # (the shape rule picks the scan here: one head, degree order)
    got, E, S = ops.gat_aggregate(A, Wh, att, relu=True, want_edge_outputs=True, fill_dead_rows=False)
    csr = (A.rowptr.cpu().numpy(), A.col.cpu().numpy(), A.val.float().cpu().numpy())
    want, wE, wS = oracle.gat_f64(1, csr, Wh.float().cpu().numpy(), att.float().cpu().numpy(), 0.2)
    np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=1e-2, atol=2e-3)
    np.testing.assert_allclose(E.cpu().numpy(), wE, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(S.cpu().numpy(), wS, rtol=4e-3, atol=1e-5)
