"""The bench workload at its full size (S-100M: 4.2 M nodes, ~104 M edges, hidden 64, sparse
first layer) checked through properties that do not need a CPU pass over the whole graph:

  * sampled rows against the oracle (the rows' edges and the table rows they touch are copied out);
  * a checksum of checksums: column sums of D against sums formed edge by edge with torch ops;
  * constant table -> row sums of A;  linearity in the table;  ReLU = clamp of the linear result;
  * the same bits on a second run, with and without the row schedule;
  * both layers of the forward pass end to end on sampled rows.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda")


@pytest.fixture(scope="module")
def workload():
    import bench
    from sgracex1_amd import graphs, ops
    wl = bench.WORKLOADS["s100m"]
    A, X, W1t, W2t = bench.make_inputs(torch, graphs, ops, wl, 0, 1, dev)
    A.plan
    X.plan
    yield A, X, W1t, W2t
    del A, X
    torch.cuda.empty_cache()


from _fixtures import sample_rows as _sample_rows  # noqa: E402


def test_full_size_aggregation_properties(workload, oracle):
    from sgracex1_amd import ops
    A, _X, _W1t, _W2t = workload
    n, P = A.n_rows, 64
    assert n == 1 << 22 and A.nnz > 100_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    H = (torch.rand((n, P), generator=g, device=dev) - 0.25).half()
    D = ops.spmm(A, H, relu=False)
    # same bits on a second run and without the row schedule
    assert torch.equal(ops.spmm(A, H, relu=False), D)
    assert torch.equal(ops.spmm(A, H, relu=False, use_plan=False), D)
    # ReLU is the clamp of the linear result (K.cpp:2586-2590)
    Dr = ops.spmm(A, H, relu=True)
    assert torch.equal(Dr, torch.where(D > 0, D, torch.zeros_like(D)))

    # sampled rows against the oracle
    gr = torch.Generator(device=dev)
    gr.manual_seed(6)
    rows = torch.randint(0, n, (4096,), generator=gr, device=dev).sort().values
    srp, scol, sval, uniq = _sample_rows(A, rows)
    table = H[uniq].float().cpu().numpy()
    want = oracle.spmm_f32(0, (srp, scol, sval), table)
    got = D[rows].float().cpu().numpy()
    # fp32 sums of <= a few dozen terms, one rounding to fp16: within one half-ulp step of the oracle's fp32 result
    want16 = want.astype(np.float16).astype(np.float32)
    assert np.mean(got == want16) > 0.999
    np.testing.assert_allclose(got, want, rtol=1.5e-3, atol=1e-4)

    # checksum of checksums: column sums of D against edge-by-edge sums formed by torch
    col_sum = torch.zeros(P, dtype=torch.float64, device=dev)
    step = 8_000_000
    for e0 in range(0, A.nnz, step):
        e1 = min(A.nnz, e0 + step)
        col_sum += (A.val[e0:e1].double()[:, None] * H[A.col[e0:e1].long()].double()).sum(0)
    got_sum = D.double().sum(0)
    # each D element carries one fp16 rounding (rel 2^-11, zero mean); the sum of 4.2 M of them
    scale = D.double().abs().sum(0)
    assert ((got_sum - col_sum).abs() <= scale * 2.0 ** -11 * 0.02 + 1e-6).all(), (got_sum - col_sum, scale)

    # constant table -> row sums of A
    ones = torch.ones((n, 8), dtype=torch.float16, device=dev)
    rs = ops.spmm(A, ones, relu=False)
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    want_rs = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, row, A.val.double())
    assert (rs.double() - want_rs[:, None]).abs().max() <= 2.0 ** -10 * float(want_rs.max())
    assert torch.equal(rs[:, 0], rs[:, 7])

    # linearity in the table (fp32 tensors: the only roundings left are those of the sums)
    A32 = ops.Csr(A.rowptr, A.col, A.val.float(), A.n_cols, A._plan)
    H1 = torch.rand((n, 16), generator=g, device=dev)
    H2 = torch.rand((n, 16), generator=g, device=dev)
    lhs = ops.spmm(A32, H1 + 2 * H2, relu=False)
    rhs = ops.spmm(A32, H1, relu=False) + 2 * ops.spmm(A32, H2, relu=False)
    assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1e-6)


def test_full_size_two_layer_forward_on_sampled_rows(workload, oracle):
    """Both layers of the bench step; the sampled output rows are recomputed by the oracle from the
    GPU's own intermediate tables (H1 = X.W1, D1, H2 = D1.W2), each stage on its own."""
    from sgracex1_amd import ops
    A, X, W1t, W2t = workload
    n = A.n_rows
    D1 = ops.layer_forward(A, X, W1t, relu=True)
    D2 = ops.layer_forward(A, D1, W2t, relu=False)
    gr = torch.Generator(device=dev)
    gr.manual_seed(9)
    rows = torch.randint(0, n, (2048,), generator=gr, device=dev).sort().values

    # stage 1 on the sampled rows: sparse X . W1
    xrp, xcol, xval, xuniq = _sample_rows(X, rows)
    W1 = ops.transpose(W1t)                                   # [F_in, hidden]
    H1 = ops.spmm(X, W1, relu=False)
    want_h1 = oracle.spmm_f32(0, (xrp, xcol, xval), W1[xuniq].float().cpu().numpy())
    np.testing.assert_allclose(H1[rows].float().cpu().numpy(), want_h1, rtol=2e-3, atol=2e-4)
    # stage 2: A . H1 with ReLU
    arp, acol, aval, auniq = _sample_rows(A, rows)
    want_d1 = oracle.spmm_f32(1, (arp, acol, aval), H1[auniq].float().cpu().numpy())
    np.testing.assert_allclose(D1[rows].float().cpu().numpy(), want_d1, rtol=2e-3, atol=2e-4)
    # stage 3: dense D1 . W2
    H2 = ops.xw_dense(D1, W2t)
    want_h2 = D1[rows].float().cpu().numpy() @ W2t.float().cpu().numpy().T
    np.testing.assert_allclose(H2[rows].float().cpu().numpy(), want_h2, rtol=2e-3, atol=3e-4)
    # stage 4: A . H2
    want_d2 = oracle.spmm_f32(0, (arp, acol, aval), H2[auniq].float().cpu().numpy())
    np.testing.assert_allclose(D2[rows].float().cpu().numpy(), want_d2, rtol=2e-3, atol=3e-4)
    assert torch.isfinite(D2.float()).all() and float(D2.float().abs().max()) > 0


def test_full_size_exact_mode_bits(workload, oracle):
    """SGX_ACC_REF_HALF at the bench's full size: sampled sblocks (4 consecutive rows starting at a multiple
    of 4, SPMM_BLOCK = 4) recomputed by the oracle's model of the reference's half arithmetic -- same bits."""
    from sgracex1_amd import ops
    A, _X, _W1t, _W2t = workload
    n, P = A.n_rows, 64
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    H = (torch.rand((n, P), generator=g, device=dev) - 0.3).half()
    D = ops.spmm(A, H, relu=True, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4)
    starts = (torch.randint(0, n // 4, (400,), generator=g, device=dev).unique() * 4)
    rows = (starts[:, None] + torch.arange(4, device=dev)[None, :]).reshape(-1)
    srp, scol, sval, uniq = _sample_rows(A, rows)
    table = H[uniq].cpu().numpy()                                  # float16 [n_uniq, P]
    eye = np.eye(P, dtype=np.float16)
    want = oracle.layer_refhalf(1, 1, (srp, scol, sval.astype(np.float16)), table, eye, N=rows.numel(), M_adj=table.shape[0],
                                spmm_block=4)
    got = D[rows].cpu().numpy()
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    # and it is a different, observable arithmetic: not the fp32-accumulate result everywhere
    plain = ops.spmm(A, H, relu=True)[rows].cpu().numpy()
    assert (plain.view(np.uint16) != got.view(np.uint16)).mean() > 0.05
    assert np.abs(plain.astype(np.float32) - got.astype(np.float32)).max() < 4e-3


def test_full_size_rmat_aggregation_properties(oracle):
    """SURVEY 8d's second generator at full size (R-MAT, 4.2 M nodes, ~100 M edges): the plan cuts hub rows into
    tasks and schedules the rest in degree order.  Sampled rows -- the hubs among them -- against the oracle, the
    column-sum checksum, constant table -> row sums, and the scheduled result against the unscheduled kernel."""
    import bench
    from sgracex1_amd import graphs, ops
    wl = bench.WORKLOADS["s100m-rmat"]
    A, _X, _W1t, _W2t = bench.make_inputs(torch, graphs, ops, wl, 0, 1, dev)
    del _X
    torch.cuda.empty_cache()
    n, P = A.n_rows, 64
    plan = A.plan
    assert plan.long_rows > 100 and plan.reordered
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    H = (torch.rand((n, P), generator=g, device=dev) - 0.25).half()
    D = ops.spmm(A, H, relu=False)
    assert torch.equal(ops.spmm(A, H, relu=False), D)                      # reproducible: no atomics on the split path
    # sampled rows: 2048 random ones and the 64 longest
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    rows = torch.cat([torch.randint(0, n, (2048,), generator=g, device=dev), torch.topk(deg, 64).indices]).unique()
    assert int(deg[rows].max()) > plan.long_threshold
    srp, scol, sval, uniq = _sample_rows(A, rows)
    want = oracle.spmm_f32(0, (srp, scol, sval), H[uniq].float().cpu().numpy())
    got = D[rows].float().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)
    assert np.mean(got == want.astype(np.float16).astype(np.float32)) > 0.99
    # checksum of checksums: column sums of D against edge-by-edge sums formed by torch
    col_sum = torch.zeros(P, dtype=torch.float64, device=dev)
    step = 8_000_000
    for e0 in range(0, A.nnz, step):
        e1 = min(A.nnz, e0 + step)
        col_sum += (A.val[e0:e1].double()[:, None] * H[A.col[e0:e1].long()].double()).sum(0)
    scale = D.double().abs().sum(0)
    assert ((D.double().sum(0) - col_sum).abs() <= scale * 2.0 ** -11 * 0.02 + 1e-6).all()
    # the schedule changes no sum of a short row; hub rows are summed in 4096-edge tasks: equal to fp32 rounding
    plain = ops.spmm(A, H, relu=False, use_plan=False)
    short = deg <= plan.long_threshold
    assert torch.equal(plain[short], D[short])
    assert torch.allclose(plain[~short].float(), D[~short].float(), rtol=2e-3, atol=2e-4)
    # constant table -> row sums of A
    ones = torch.ones((n, 8), dtype=torch.float16, device=dev)
    rs = ops.spmm(A, ones, relu=False)
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    want_rs = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, row, A.val.double())
    assert (rs.double() - want_rs[:, None]).abs().max() <= 2.0 ** -10 * float(want_rs.max()) + 1e-3


def test_full_size_exact_mode_dense_stage_bits(workload, oracle):
    """SGX_ACC_REF_HALF on the dense X.W stage at the bench's size (4.2 M rows, 64 -> 64, the lane-group kernel with
    the LDS weight tile): sampled sblocks recomputed by the oracle's model -- an identity adjacency passes H through
    its second stage unchanged (1.0 * h and additions of +0 are exact) -- same bits."""
    from sgracex1_amd import ops
    A, _X, _W1t, W2t = workload
    n = A.n_rows
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    D1 = (torch.rand((n, 64), generator=g, device=dev) - 0.3).half()
    H2 = ops.xw_dense(D1, W2t, acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4)
    starts = torch.randint(0, n // 4, (300,), generator=g, device=dev).unique() * 4
    starts = torch.cat([starts, torch.tensor([0, n - 4], device=dev)]).unique()
    rows = (starts[:, None] + torch.arange(4, device=dev)[None, :]).reshape(-1)
    m = rows.numel()
    eye = (np.arange(m + 1, dtype=np.int32), np.arange(m, dtype=np.int32), np.ones(m, dtype=np.float16))
    want = oracle.layer_refhalf(1, 0, eye, D1[rows].cpu().numpy(), W2t.cpu().numpy(), N=m, M_adj=m, spmm_block=4)
    got = H2[rows].cpu().numpy()
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    plain = ops.xw_dense(D1, W2t)[rows].cpu().numpy()
    assert (plain.view(np.uint16) != got.view(np.uint16)).mean() > 0.02      # an observable arithmetic, as in the CSR stage


def test_full_size_rmat_gat_aggregate_properties():
    """The GAT aggregate on the full-size power-law graph (4.2 M nodes, ~100 M edges, hubs of 10^5 entries through tasks,
    degree order, the one-piece tail), one head of 64 columns -- **parity unpinned** like all of GAT; properties that need
    no CPU pass: a table whose rows are all the same vector comes back as that vector on every row with a live entry
    (softmax weights sum to 1, whatever the scores) and as 0 elsewhere; the one-walk form (csrc/gat_fused.hip) and the
    two stages (scores gathered per edge, weights through memory) agree on every row of a random table; the same bits on
    a second run."""
    import bench
    from sgracex1_amd import _lib, graphs, ops
    wl = bench.WORKLOADS["s100m-rmat"]
    A, _X, _W1t, _W2t = bench.make_inputs(torch, graphs, ops, wl, 0, 1, dev)
    del _X
    torch.cuda.empty_cache()
    n, P = A.n_rows, 64
    assert A.gat_plan.long_rows > 1000 and A.gat_plan.reordered
    g = torch.Generator(device=dev)
    g.manual_seed(31)
    att = ((torch.rand(2 * P, generator=g, device=dev) * 2 - 1) * 0.3).half()
    deg = (A.rowptr[1:] - A.rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    # (the generator gives every node a self loop: some rows lose all their entries to the mask here, a hub among them)
    masked = torch.zeros(n, dtype=torch.bool, device=dev)
    masked[torch.arange(5, n, 1001, device=dev)] = True
    masked[torch.topk(deg, 3).indices[-1]] = True
    val = torch.where(masked[row], torch.zeros_like(A.val), A.val)
    A = ops.Csr(A.rowptr, A.col, val, A.n_cols, A.plan)
    live = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row, (val.float() > 0).long()) > 0
    assert torch.equal(live, ~masked)
    del row, val
    v = (torch.rand(P, generator=g, device=dev) * 2 - 1).half()
    same = v.expand(n, P).contiguous()
    D = ops.gat_aggregate(A, same, att, relu=False, fill_dead_rows=False)
    assert (D[live].float() - v.float()).abs().max() <= 2.0 ** -10 * float(v.float().abs().max())     # one binary16 ulp of the largest entry
    assert not D[~live].any() and int((~live).sum()) > 0
    del same, D
    Wh = (torch.rand((n, P), generator=g, device=dev) - 0.4).half()
    one_walk = ops.gat_aggregate(A, Wh, att, relu=True, fill_dead_rows=False)
    assert torch.equal(ops.gat_aggregate(A, Wh, att, relu=True, fill_dead_rows=False), one_walk)
    with _lib.tuning(SGX_GAT_FUSED="0"):
        two_stage = ops.gat_aggregate(A, Wh, att, relu=True, fill_dead_rows=False)
    assert not torch.equal(one_walk, two_stage)                                 # (two forms ran)
    torch.testing.assert_close(one_walk.float(), two_stage.float(), rtol=2e-3, atol=1e-3)
    hubs = torch.topk(deg, 64).indices
    assert int(deg[hubs].min()) > 1000 and torch.isfinite(one_walk[hubs].float()).all()
