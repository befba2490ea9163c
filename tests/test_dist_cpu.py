"""Multi-GPU path, rehearsed on the CPU with gloo (world_size 2 and 3): row partition, the halo
index construction (bit-exact column remap), the all-gather and all-to-all exchanges.  The
local compute is injected (the oracle's CSR product on CPU tensors) -- on the GPU box the same
code runs with the HIP backend over RCCL."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(n, avg_deg, seed):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, n)
    deg[rng.random(n) < 0.1] = 0
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    ci = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in deg] + [np.zeros(0, np.int64)]).astype(np.int32)
    va = rng.standard_normal(rp[-1]).astype(np.float32)
    return rp, ci, va


def _worker(rank, world, port, tmp, balance_nnz):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from sgracex1_amd import dist as D

        n, m, p = 257, 19, 12
        rp, ci, va = _graph(n, 5.0, 3)
        rng = np.random.default_rng(4)
        X = rng.standard_normal((n, m)).astype(np.float32)
        W = rng.standard_normal((m, p)).astype(np.float32)
        Wt = torch.as_tensor(np.ascontiguousarray(W.T))
        want = O.layer_f64(1, 1, (rp, ci, va), X, np.ascontiguousarray(W.T), h_round=1)

        class LocalCsr:                      # what ops.Csr is on the GPU
            def __init__(self, rowptr, col, val, n_cols):
                self.rowptr, self.col, self.val, self.n_cols = rowptr, col, val, n_cols

        def xw(fea, Wt_):
            return fea @ Wt_.t()

        def spmm(adj, table, relu):
            out = O.spmm_f32(int(relu), (adj.rowptr.numpy(), adj.col.numpy(), adj.val.numpy()), table.numpy())
            return torch.as_tensor(out)

        def spmm_partial(adj, table):
            return spmm(adj, table, False)

        def spmm_finish(adj, table, partial, relu):
            out = partial + spmm(adj, table, False)
            return torch.clamp(out, min=0) if relu else out

        def gat(adj, table, att, alpha, relu, fill_row=None, n_nodes=None):    # SG.py:634-661 on the rank's rows, dense
            n_loc, n_tab = adj.rowptr.numel() - 1, table.shape[0]
            deg = (adj.rowptr[1:] - adj.rowptr[:-1]).long()
            rows = torch.repeat_interleave(torch.arange(n_loc), deg)
            mask = torch.zeros((n_loc, n_tab))
            mask[rows, adj.col.long()] = (adj.val > 0).float()
            Pn = table.shape[1]
            e = torch.nn.functional.leaky_relu((table[:n_loc] @ att[:Pn])[:, None] + (table @ att[Pn:])[None, :], alpha)
            a = torch.softmax(torch.where(mask > 0, e, torch.full_like(e, -9e15)), dim=1)
            dead = mask.sum(1, keepdim=True) == 0
            a = torch.where(dead, torch.zeros_like(a), a)
            out = a @ table
            if fill_row is not None:                              # the mean row of ALL nodes, reduced by layer_halo
                assert n_nodes == n
                out = torch.where(dead, fill_row[None, :].to(out.dtype), out)
            return torch.clamp(out, min=0) if relu else out

        def xw_act(z, Wt_, relu):
            out = z @ Wt_.t()
            return torch.clamp(out, min=0) if relu else out

        backend = D.Backend(xw=xw, spmm=spmm, spmm_partial=spmm_partial, spmm_finish=spmm_finish, gat=gat, xw_act=xw_act,
                            col_sums=lambda h: h.double().sum(0).float())
        trp, tci, tva = torch.as_tensor(rp), torch.as_tensor(ci), torch.as_tensor(va)
        bounds = D.row_partition(n, world, trp if balance_nnz else None)
        assert bounds[0] == 0 and bounds[-1] == n and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        lo, hi = bounds[rank], bounds[rank + 1]
        lrp, lci, lva = D.slice_rows(trp, tci, tva, lo, hi)
        assert int(lrp[0]) == 0 and int(lrp[-1]) == lci.numel()
        Xl = torch.as_tensor(X[lo:hi])

        # exchange (i): all-gather of H, global column indices
        d1 = D.layer_allgather(backend, LocalCsr(lrp, lci, lva, n), Xl, Wt, True, bounds)
        np.testing.assert_allclose(d1.numpy(), want[lo:hi], rtol=1e-5, atol=1e-5)

        # exchange (ii): halo rows only; the remap must be bit exact
        plan = D.build_halo_plan(lci, bounds, rank)
        H_full = torch.as_tensor(X) @ Wt.t()
        table = torch.empty((plan.n_table, p))
        table[:plan.n_own] = H_full[lo:hi]
        off = plan.n_own
        owner = np.searchsorted(np.array(bounds), lci.numpy(), side="right") - 1
        for g in range(world):
            if g == rank:
                continue
            rows = np.unique(lci.numpy()[owner == g])
            assert plan.recv_counts[g] == len(rows)
            table[off:off + len(rows)] = H_full[rows]
            off += len(rows)
        assert off == plan.n_table
        assert torch.equal(table[plan.col_compact.long()], H_full[lci.long()])      # same row behind every edge
        assert plan.col_compact.dtype == torch.int32
        assert int(plan.col_compact.max()) < plan.n_table if lci.numel() else True
        d2 = D.layer_halo(backend, LocalCsr(lrp, plan.col_compact, lva, plan.n_table), Xl, Wt, True, plan)
        assert torch.equal(d1, d2)                                                   # same sums, same order
        # exchange (iii): the halo rows travel while the own-partition edges are summed
        own, halo = D.split_own_halo(lrp, plan.col_compact, lva, plan.n_own)
        assert int(own[0][-1]) + int(halo[0][-1]) == lci.numel()
        assert (own[1] < plan.n_own).all() and (halo[1] < sum(plan.recv_counts)).all() if lci.numel() else True
        d3 = D.layer_halo_overlap(backend, LocalCsr(*own, plan.n_own), LocalCsr(*halo, sum(plan.recv_counts)), Xl, Wt,
                                  True, plan)
        np.testing.assert_allclose(d3.numpy(), want[lo:hi], rtol=1e-5, atol=1e-5)
        # aggregate-first order: the rows of X travel instead of the rows of H; act((A.X).W) on every exchange
        for d_swapped in (
                D.layer_allgather(backend, LocalCsr(lrp, lci, lva, n), Xl, Wt, True, bounds, aggregate_first=True),
                D.layer_halo(backend, LocalCsr(lrp, plan.col_compact, lva, plan.n_table), Xl, Wt, True, plan,
                             aggregate_first=True),
                D.layer_halo_overlap(backend, LocalCsr(*own, plan.n_own), LocalCsr(*halo, sum(plan.recv_counts)), Xl, Wt,
                                     True, plan, aggregate_first=True)):
            assert d_swapped.shape == (hi - lo, p)
            np.testing.assert_allclose(d_swapped.numpy(), want[lo:hi], rtol=1e-4, atol=1e-4)
        with pytest.raises(ValueError):
            D.layer_halo(backend, LocalCsr(lrp, plan.col_compact, lva, plan.n_table), Xl, Wt, True, plan,
                         attention=torch.zeros(2 * p), aggregate_first=True)
        # GAT over the same halo rows against the single-process dense formula
        att = torch.as_tensor(rng.standard_normal(2 * p).astype(np.float32)) * 0.3
        H_all = H_full
        pos = torch.zeros((n, n))
        deg_all = np.diff(rp)
        pos[np.repeat(np.arange(n), deg_all), ci.astype(np.int64)] = torch.as_tensor((va > 0).astype(np.float32))
        e_all = torch.nn.functional.leaky_relu((H_all @ att[:p])[:, None] + (H_all @ att[p:])[None, :], 0.2)
        # the reference's formula as it stands (SG.py:638-641): a row without a positive entry keeps -9e15 everywhere,
        # its softmax is uniform over ALL n nodes and it receives the mean of all rows of Wh -- also when the rows of
        # Wh live on other ranks (rows without any edge are 10 % of this graph, and every rank holds some)
        a_all = torch.softmax(torch.where(pos > 0, e_all, torch.full_like(e_all, -9e15)), dim=1)
        dead_all = pos.sum(1) == 0
        assert bool(dead_all[lo:hi].any()) and torch.allclose(a_all[dead_all], torch.full_like(a_all[dead_all], 1.0 / n))
        want_gat = torch.clamp(a_all @ H_all, min=0)
        adj_gat = LocalCsr(lrp, plan.col_compact, lva, plan.n_table)
        d4 = D.layer_halo(backend, adj_gat, Xl, Wt, True, plan, attention=att)
        np.testing.assert_allclose(d4.numpy(), want_gat[lo:hi].numpy(), rtol=1e-4, atol=1e-5)
        assert D.any_rank_has_dead_rows(adj_gat) is True        # decided once per adjacency, by all ranks together
        # the same PLAN with an adjacency whose rows all keep a live edge (every stored value positive, a self loop added
        # to the empty rows): the decision follows the adjacency's values -- a flag kept on the plan would be stale here
        live_val = lva.abs() + 0.5
        adj_live = LocalCsr(lrp, plan.col_compact, live_val, plan.n_table)
        adj_live.has_dead_rows = bool(((lrp[1:] - lrp[:-1]) == 0).any())
        expect_any = torch.tensor([float(adj_live.has_dead_rows)])
        dist.all_reduce(expect_any)
        assert D.any_rank_has_dead_rows(adj_live) is bool(expect_any.item() > 0)
        # an in-place change of the values asks again (all ranks together)
        adj_flip = LocalCsr(lrp, plan.col_compact, lva.clone(), plan.n_table)
        adj_flip.has_dead_rows = False
        assert D.any_rank_has_dead_rows(adj_flip) is False
        adj_flip.val.mul_(1.0)
        adj_flip.has_dead_rows = True
        assert D.any_rank_has_dead_rows(adj_flip) is True
        # the cheaper sparse answer stays available: such rows give 0
        d5 = D.layer_halo(backend, LocalCsr(lrp, plan.col_compact, lva, plan.n_table), Xl, Wt, True, plan, attention=att,
                          fill_dead_rows=False)
        assert not d5[dead_all[lo:hi]].any() and torch.allclose(d5[~dead_all[lo:hi]], d4[~dead_all[lo:hi]])
        # the all-gather as one batch of point-to-point transfers: the same table, the same result
        d6 = D.layer_allgather(backend, LocalCsr(lrp, lci, lva, n), Xl, Wt, True, bounds, direct=True)
        assert torch.equal(d6, d1)
        # bytes moved: the halo exchange never receives more rows than the all-gather would
        assert sum(plan.recv_counts) <= n - (hi - lo)
        sent = torch.tensor([sum(plan.send_counts)], dtype=torch.int64)
        recv = torch.tensor([sum(plan.recv_counts)], dtype=torch.int64)
        dist.all_reduce(sent)
        dist.all_reduce(recv)
        assert int(sent) == int(recv)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balance_nnz", [(2, False), (2, True), (3, True), (4, False), (8, True)])
def test_partitioned_layer_matches_single_process(world, balance_nnz):
    from oracle import oracle as O
    O.build()
    port = 29500 + (os.getpid() % 2000) + world * 7 + int(balance_nnz)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, balance_nnz), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(tmp, f"ok{r}")) for r in range(world))
