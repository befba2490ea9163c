"""The one-step tail of a degree-ordered plan -- rows of at most 8 edges, 64 to a wavefront (spmm_short_rows in
csrc/spmm_csr.hip; the same walk in the GAT aggregate's second stage, csrc/gat.hip) -- against the sblock walk of the
same rows (SGX_SPMM_NO_SHORT_TAIL): the same fma chain per output element, so the same BITS, for every lane split, element
type and entry point that takes the path, and against the oracle on sampled rows.  The reference groups rows per pipelined
loop by a build constant (SPMM_BLOCK, K.cpp:826-845); this is the same idea with the group chosen by row length."""
import numpy as np
import pytest
import torch

from _fixtures import sample_rows

pytestmark = pytest.mark.gpu
dev = torch.device("cuda")


def _powerlaw_graph(n, seed, dtype):
    """mostly one-step rows (many self-loop-only and empty ones), a tail of long rows and a few hubs: the plan orders it"""
    from sgracex1_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    deg = (torch.rand(n, generator=g, device=dev) ** -1.1).clamp(max=3000).long()
    deg[torch.rand(n, generator=g, device=dev) < 0.1] = 0
    deg[:3] = torch.tensor([9000, 5000, 8], device=dev)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(deg, 0)
    nnz = int(rp[-1])
    col = torch.randint(0, n, (nnz,), generator=g, device=dev, dtype=torch.int32)
    val = (torch.rand(nnz, generator=g, device=dev) * 2 - 0.7).to(dtype)
    A = ops.Csr(rp.to(torch.int32), col, val, n)
    assert A.plan.reordered and A.plan.long_rows > 0
    return A


@pytest.mark.parametrize("dtype,P", [(torch.float16, 64), (torch.float16, 128), (torch.float16, 256), (torch.float16, 100),
                                     (torch.float32, 64), (torch.float32, 32), (torch.float16, 512)])
def test_short_tail_gives_the_bits_of_the_sblock_walk(oracle, dtype, P):
    from sgracex1_amd import _lib, ops
    n = 300_000
    A = _powerlaw_graph(n, 17 + P, dtype)
    g = torch.Generator(device=dev)
    g.manual_seed(P)
    H = (torch.rand((n, P), generator=g, device=dev) - 0.4).to(dtype)
    got = ops.spmm(A, H, relu=True)
    with _lib.tuning(SGX_SPMM_NO_SHORT_TAIL="1"):
        ref = ops.spmm(A, H, relu=True)
    assert torch.equal(got, ref)
    assert torch.equal(got, ops.spmm(A, H, relu=True))                    # and run to run
    # two-pass aggregation (the multi-GPU halo overlap): fp32 partial sums out, then in
    part = ops.spmm_acc(A, H, partial_out=True)
    fin = ops.spmm_acc(A, H, relu=True, acc_in=part)
    with _lib.tuning(SGX_SPMM_NO_SHORT_TAIL="1"):
        part0 = ops.spmm_acc(A, H, partial_out=True)
        fin0 = ops.spmm_acc(A, H, relu=True, acc_in=part0)
    assert torch.equal(part, part0) and torch.equal(fin, fin0)
    # sampled rows against exact math: one-edge rows, empty rows, 8-edge rows, the hubs
    deg = A.rowptr.diff()
    pick = torch.cat([torch.nonzero(deg == 0).flatten()[:20], torch.nonzero(deg == 1).flatten()[:100],
                      torch.nonzero(deg == 8).flatten()[:40], torch.nonzero((deg > 1) & (deg < 8)).flatten()[:100],
                      torch.tensor([0, 1, 2], device=dev)])
    srp, scol, sval, uniq = sample_rows(A, pick)
    table = H[uniq].float().cpu().numpy()
    eye = (np.arange(len(pick) + 1, dtype=np.int32), np.arange(len(pick), dtype=np.int32), np.ones(len(pick), np.float32))
    want = oracle.layer_f64(0, 0, eye, (srp, scol, sval), np.ascontiguousarray(table.T), h_round=2 if dtype == torch.float16 else 1)
    want = np.maximum(want, 0)
    tol = dict(rtol=1e-2, atol=2e-3) if dtype == torch.float16 else dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got[pick].float().cpu().numpy(), want, **tol)
    assert not got[deg == 0].any()


@pytest.mark.parametrize("heads", [1, 8])
def test_gat_aggregate_short_tail_gives_the_same_bits(heads):
    """the edge-softmax aggregate's second stage walks the same degree order; E and S are stage A's and do not depend on it"""
    from sgracex1_amd import _lib, ops
    n, P = 200_000, 64
    A = _powerlaw_graph(n, 5 + heads, torch.float16)
    assert A.gat_plan.reordered
    g = torch.Generator(device=dev)
    g.manual_seed(heads)
    Wh = (torch.rand((n, P), generator=g, device=dev) - 0.5).half()
    att = ((torch.rand(2 * P, generator=g, device=dev) * 2 - 1) * 0.3).half()
    got, E, S = ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, want_edge_outputs=True)
    with _lib.tuning(SGX_SPMM_NO_SHORT_TAIL="1"):
        ref, E0, S0 = ops.gat_aggregate(A, Wh, att, relu=True, heads=heads, want_edge_outputs=True)
    assert torch.equal(got, ref) and torch.equal(E, E0) and torch.equal(S, S0)
    assert torch.isfinite(got.float()).all() and got.abs().max() > 0
