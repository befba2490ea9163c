"""The row schedule (sgx_plan) built on the device against the rules restated in numpy: which rows are cut into
tasks and where, the lane-group utilisation of the natural order, and -- when that is low -- the degree order of the
short rows (stable: longest first, ascending row id inside a step count).  The reference makes the matching decisions
on the host (K.cpp:3517-3523 row split over threads, :826-845 rows per pipelined loop); the plan is this build's form
of them, so what is pinned here is that the device builder produces exactly the arrays the rules define."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda")


def default_cut(nnz):
    cut = 64
    while cut < 4096 and (2 * cut) * (2 * cut) * 4 <= nnz:
        cut *= 2
    return cut


def plan_by_rule(rowptr, long_threshold=0, chunk=0, reorder_below=0.7):
    rp = np.asarray(rowptr, np.int64)
    n = rp.size - 1
    nnz = int(rp[-1])
    small = nnz < (1 << 20)
    thr = 64 if small else default_cut(nnz)
    ch = 64 if small else thr
    if not small and long_threshold >= 8:
        thr = long_threshold // 8 * 8
        ch = chunk // 8 * 8 if chunk >= 8 else thr
    deg = rp[1:] - rp[:-1]
    is_long = deg > thr
    long_row = np.nonzero(is_long)[0]
    tasks = (deg[long_row] + ch - 1) // ch
    long_first = np.concatenate([[0], np.cumsum(tasks)])
    task_row = np.repeat(long_row, tasks)
    k = np.arange(task_row.size) - np.repeat(long_first[:-1], tasks)
    task_e0 = rp[task_row] + k * ch
    task_e1 = np.minimum(task_e0 + ch, rp[task_row + 1])
    steps = np.where(is_long, 0, (deg + 7) // 8)
    pad = (-n) % 8
    groups = np.concatenate([steps, np.zeros(pad, np.int64)]).reshape(-1, 8)
    useful, spent = int(steps.sum()), int(groups.max(axis=1).sum() * 8)
    util = useful / spent if spent else 1.0
    order = None
    if util < reorder_below and n - long_row.size > 0:
        short = np.nonzero(~is_long)[0]
        order = short[np.argsort(-((deg[short] + 7) // 8), kind="stable")]
    return dict(long_threshold=thr, long_row=long_row, long_first=long_first if long_row.size else np.zeros(0, np.int64),
                task_row=task_row, task_e0=task_e0, task_e1=task_e1, row_order=order, utilization=util)


def _rowptr(deg):
    rp = np.zeros(deg.size + 1, np.int64)
    np.cumsum(deg, out=rp[1:])
    assert rp[-1] < 2 ** 31
    return rp


def _compare(rp, **kw):
    from sgracex1_amd import ops
    plan = ops.Plan(torch.as_tensor(rp.astype(np.int32), device=dev), **kw)
    want = plan_by_rule(rp, **kw)
    assert plan.long_threshold == want["long_threshold"]
    assert plan.long_rows == want["long_row"].size
    assert abs(plan.natural_utilization - want["utilization"]) < 1e-6
    for name in ("long_row", "long_first", "task_row", "task_e0", "task_e1"):
        got = plan.export(name).cpu().numpy()
        np.testing.assert_array_equal(got, want[name], err_msg=name)
    got_order = plan.export("row_order").cpu().numpy()
    assert plan.reordered == (want["row_order"] is not None)
    if want["row_order"] is not None:
        np.testing.assert_array_equal(got_order, want["row_order"])
    else:
        assert got_order.size == 0
    return plan


def _degrees(kind, n, rng):
    if kind == "uniform":
        return rng.poisson(18, n)
    if kind == "powerlaw":
        return np.minimum((rng.pareto(1.1, n) * 3).astype(np.int64), 200_000)
    if kind == "empty":
        return np.zeros(n, np.int64)
    if kind == "hubs":                                    # a handful of very long rows among short ones, the last row included
        d = rng.integers(0, 12, n)
        d[rng.integers(0, n, 7)] = rng.integers(5_000, 90_000, 7)
        d[-1] = 70_001
        d[0] = 65
        return d
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "powerlaw", "empty", "hubs"])
@pytest.mark.parametrize("n", [1, 7, 8, 257, 4096, 4097, 70_003, 1_500_001])
def test_plan_matches_rules(kind, n):
    rng = np.random.default_rng(n * 31 + len(kind))
    _compare(_rowptr(_degrees(kind, n, rng)))


def test_plan_many_row_blocks_and_callers_cut():
    """more rows than 1024 blocks of 4096 (the row blocks double), and cuts chosen by the caller -- 256 / 256 is the GAT
    aggregate's, 1000 / 500 rounds down to multiples of 8, a chunk of 0 follows the threshold"""
    rng = np.random.default_rng(5)
    deg = np.minimum((rng.pareto(1.3, 4_300_000) * 2).astype(np.int64), 30_000)
    rp = _rowptr(deg)
    plan = _compare(rp)
    assert plan.long_rows > 0 and plan.reordered
    _compare(rp, long_threshold=256, chunk=256)
    _compare(rp, long_threshold=1000, chunk=500)
    _compare(rp, long_threshold=2048, chunk=0)


def test_plan_no_rows():
    from sgracex1_amd import ops
    plan = ops.Plan(torch.zeros(1, dtype=torch.int32, device=dev))
    assert plan.long_rows == 0 and not plan.reordered and plan.export("row_order").numel() == 0


def test_plan_drives_the_same_sums():
    """a graph whose plan has tasks AND the degree order: the aggregation under it equals the unplanned walk bit for bit
    where no row is cut (same fma chain), and to fp32 rounding where rows are (partial sums merged in task order)"""
    from sgracex1_amd import graphs, ops
    A = graphs.rmat_graph(18, 6_000_000, seed=11)
    assert A.plan.long_rows > 0 and A.plan.reordered
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    H = (torch.rand((A.n_cols, 64), generator=gen, device=dev) - 0.5).half()
    with_plan = ops.spmm(A, H, relu=False)
    without = ops.spmm(A, H, relu=False, use_plan=False)
    cut = torch.zeros(A.n_rows, dtype=torch.bool, device=dev)
    cut[A.plan.export("long_row").long()] = True
    assert torch.equal(with_plan[~cut], without[~cut])
    torch.testing.assert_close(with_plan[cut].float(), without[cut].float(), rtol=2e-3, atol=2e-3)


def test_plan_window_order_of_the_sparse_feature_stage():
    """win_order: the rows of every 64-row window by length, longest first, ties in row order -- what the LDS form of the
    sparse X.W stage deals to its sub-tiles (xw_sparse_lds.hip).  Built for matrices of 2^20 entries and more without cut
    rows; the last window is ragged (rows past the end count as empty)."""
    from sgracex1_amd import ops
    rng = np.random.default_rng(77)
    n = 70_003
    deg = rng.poisson(18, n)
    deg[rng.integers(0, n, 500)] = 0
    rp = _rowptr(deg)
    assert rp[-1] >= 1 << 20
    plan = ops.Plan(torch.as_tensor(rp.astype(np.int32), device=dev))
    assert plan.long_rows == 0
    got = plan.export("win_order").cpu().numpy()
    n_win = (n + 63) // 64
    assert got.dtype == np.uint8 and got.size == n_win * 64
    padded = np.concatenate([deg, np.zeros(n_win * 64 - n, np.int64)]).reshape(n_win, 64)
    want = np.argsort(-padded, axis=1, kind="stable").astype(np.uint8)
    np.testing.assert_array_equal(got.reshape(n_win, 64), want)
    # no such array where the LDS form cannot run: a small matrix, a matrix with cut rows
    small = ops.Plan(torch.as_tensor(_rowptr(rng.poisson(3, 5000)).astype(np.int32), device=dev))
    assert small.export("win_order").numel() == 0
    hub = deg.copy()
    hub[5] = 200_000
    cut = ops.Plan(torch.as_tensor(_rowptr(hub).astype(np.int32), device=dev))
    assert cut.long_rows == 1 and cut.export("win_order").numel() == 0


@pytest.mark.parametrize("kind", ["uniform", "powerlaw", "empty", "hubs"])
def test_plan_entry_windows_of_the_gat_scan(kind):
    """scan_win: per boundary between windows of 64 stored entries, the first row starting at or behind it with its first
    entry, and where the window before it ends for the scan -- the same pair, or the row before and its first entry when
    that row is a long one (it belongs to the tasks).  Built for plans cut at 256 entries (Csr.gat_plan) or without a
    longer row; not for other cuts."""
    from sgracex1_amd import ops
    rng = np.random.default_rng(len(kind))
    deg = _degrees(kind, 400_003, rng).astype(np.int64)                 # (a small matrix is cut at its own threshold, whatever the caller says)
    if kind == "hubs":
        deg[[5, 6, 40_000]] = [257, 256, 1_000]
    rp = _rowptr(deg)
    rp_t = torch.as_tensor(rp, dtype=torch.int32, device=dev)
    nnz = int(rp[-1])
    plan = ops.Plan(rp_t, 256, 256)
    win = plan.export("scan_win").cpu().numpy().astype(np.int64)
    if nnz == 0:
        assert win.size == 0
        return
    assert plan.long_threshold == 256
    n_win = (nnz + 63) // 64
    assert win.size == 4 * (n_win + 1)
    win = win.reshape(n_win + 1, 4)
    e = np.minimum(np.arange(n_win + 1) * 64, nnz)
    first = np.searchsorted(rp, e, side="left")                       # first row with rowptr >= e
    assert (win[:, 0] == first).all() and (win[:, 1] == rp[first]).all()
    before = np.maximum(first - 1, 0)
    long_before = (first > 0) & (rp[first] - rp[before] > 256) & (plan.long_rows > 0)
    assert (win[:, 2] == np.where(long_before, before, first)).all()
    assert (win[:, 3] == np.where(long_before, rp[before], rp[first])).all()
    # every row of up to 256 entries lies in exactly one window's range, whole
    starts, ends = win[:-1, 1], win[1:, 3]
    assert (ends - starts <= 64 + 255).all()
    assert ops.Plan(rp_t, 512, 512).export("scan_win").numel() == 0 and ops.Plan(rp_t).export("scan_win").numel() == 0
