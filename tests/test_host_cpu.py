"""Host-side logic that needs no GPU: the pynq-shaped register map and buffers, the graph
helpers around the hot path (TU loader, batching, to_dense_adj, pooling, sym_norm2), the row
partition of the multi-GPU path."""
import os
import sys

import numpy as np
import pytest
import torch

from _fixtures import GOLD, load


@pytest.fixture(scope="module", autouse=True)
def _built():
    from sgracex1_amd import build
    build.build()


def test_allocate_and_physical_addresses():
    from sgracex1_amd import pynq_shim as P
    a = P.allocate(100, dtype=np.int32)
    b = P.allocate(shape=(4, 8), dtype=np.float16)
    assert a.shape == (100,) and a.dtype == np.int32 and not a.any()
    assert b.shape == (4, 8) and b.dtype == np.float16
    assert a.physical_address != b.physical_address
    flat, dt, win, off = P._resolve(a.physical_address)
    flat[3] = 7
    assert a[3] == 7 and dt == np.int32 and off == 0 and win is a._win
    # an address inside a buffer (the notebooks add byte offsets for D2..D4, MMN cell 31)
    flat2, _, win2, off2 = P._resolve(b.physical_address + 2 * 8 * 2)
    flat2[0] = 1.5
    assert b[2, 0] == np.float16(1.5) and off2 == 16 and win2 is b._win
    assert b[1:].physical_address == b.physical_address + 16
    a.freebuffer()
    with pytest.raises(ValueError):
        P._resolve(a.physical_address)


def test_buffer_change_tracking():
    """What decides whether a buffer's device mirror is uploaded again: a write counter that every slice assignment
    through the buffer or any of its views bumps (how the reference fills its buffers), flush(), and a content stamp for
    writes that go around the buffer object."""
    from sgracex1_amd import pynq_shim as P
    a = P.allocate(300_000, dtype=np.float16)
    v0 = a._win.version
    a[0:10] = 1.0
    assert a._win.version == v0 + 1
    view = a[100:200]
    view[:] = 2.0                                   # a view shares the buffer's counter
    assert a._win.version == v0 + 2 and a[150] == 2
    a.flush()
    assert a._win.version == v0 + 3
    s0 = P._content_stamp(np.asarray(a))
    np.asarray(a)[5] = 9                            # around the counter: the stamp of the first 4 KB moves
    assert a._win.version == v0 + 3 and P._content_stamp(np.asarray(a)) != s0
    s1 = P._content_stamp(np.asarray(a))
    np.asarray(a)[-3] = 4                           # ... and of the last 4 KB
    assert P._content_stamp(np.asarray(a)) != s1
    small = P.allocate(1000, dtype=np.int32)        # small buffers are fingerprinted whole
    s2 = P._content_stamp(np.asarray(small))
    np.asarray(small)[500] = 1
    assert P._content_stamp(np.asarray(small)) != s2
    # the pinned storage is zeroed like np.zeros, and a buffer of another dtype or shape is a view of it
    m = P.allocate(shape=(3, 5), dtype=np.int64)
    assert m.shape == (3, 5) and not m.any() and m._win.storage.numel() >= 3 * 5 * 8


def test_register_map_semantics():
    from sgracex1_amd import pynq_shim as P
    ol = P.Overlay("gnn_all.bit", device="cpu")
    ip = ol.mmult_top_0
    rm = ip.register_map
    rm.N_adj, rm.relu, rm.gemm_mode = 3327, 1, 0
    rm.D2_offset_1 = 1000 + 16 * 3327 / 1            # float arithmetic, as MMN cell 31 writes it
    assert rm.N_adj == 3327 and rm.relu == 1 and rm.D2_offset_1 == 1000 + 16 * 3327
    rm.some_new_register = 5                          # unknown names are stored like any register
    assert rm.some_new_register == 5
    assert rm.CTRL.AP_DONE == 0 and rm.CTRL.AP_IDLE == 1
    with pytest.raises(AttributeError):
        rm.never_written
    with pytest.raises(ValueError):                   # AP_START without dimensions
        rm2 = P.Overlay("gnn_all.bit", device="cpu").mmult_top_0.register_map
        rm2.CTRL.AP_START = 1
    assert not P.Overlay("gnn_all.bit", device="cpu").mmult_top_0.coo_adjacency
    assert P.Overlay("gat_all_unsigned.bit", device="cpu").mmult_top_0.coo_adjacency
    mod = P.install()
    import pynq
    assert pynq is mod and pynq.Overlay is P.Overlay and pynq.allocate is P.allocate
    del sys.modules["pynq"]


def test_mutag_loader_and_batching():
    from sgracex1_amd import pyg_lite as G
    raw = np.load(os.path.join(GOLD, "mutag_raw.npz"))
    graphs = G.load_tu_raw(raw["A"], raw["graph_indicator"], raw["graph_labels"], raw["node_labels"])
    assert len(graphs) == 188 and sum(g.num_nodes for g in graphs) == 3371
    assert sum(g.edge_index.shape[1] for g in graphs) == 7442
    # first graph of the dataset, as printed in MOL cell 4: 17 nodes, 38 edges, 7 features
    assert graphs[0].num_nodes == 17 and graphs[0].edge_index.shape[1] == 38 and graphs[0].x.shape[1] == 7
    assert sorted(set(int(g.y) for g in graphs)) == [0, 1]
    assert sum(int(g.y) for g in graphs) == 125                        # 125 positive / 63 negative
    for g in graphs[:20]:
        assert (g.x.sum(dim=1) == 1).all() and int(g.edge_index.max()) < g.num_nodes
        d = G.to_dense_adj(g.edge_index, g.num_nodes)[0]
        assert torch.equal(d, d.t()) and not torch.diagonal(d).any()     # undirected, no self loops
    b = G.collate(graphs)
    assert b.num_nodes == 3371 and b.edge_index.shape == (2, 7442) and b.num_graphs == 188   # MOL cell 10 output
    pooled = G.global_mean_pool(b.x, b.batch)
    assert pooled.shape == (188, 7) and torch.allclose(pooled.sum(dim=1), torch.ones(188))
    loader = G.DataLoader(graphs[50:100], batch_size=256)
    assert len(list(loader)) == 1


def test_sym_norm2_reproduces_the_reference_cora_values():
    """cora_adj.txt holds D^-1/2 (A + I) D^-1/2; rebuilding it from the bare structure with
    sym_norm2(fill=1) must give the stored values (text precision).  (citeseer_adj.txt was
    normalised before its duplicate edges were merged, so 4.6 % of its values cannot be rebuilt
    from the structure alone.)"""
    from sgracex1_amd.sgrace import sym_norm2
    d = load("cora")
    rp, ci, va = d["adj"]
    n = d["N"]
    row = np.repeat(np.arange(n), np.diff(rp))
    keep = row != ci
    ei = torch.as_tensor(np.stack([row[keep], ci[keep]]), dtype=torch.int64)
    ei2, norm = sym_norm2(ei, n, fill=1, dtype=torch.float32)
    assert ei2.shape[1] == len(ci)
    assert np.array_equal(ei2[0].numpy(), row) and np.array_equal(ei2[1].numpy(), ci)   # sorted by (row, col)
    np.testing.assert_allclose(norm.numpy(), va, rtol=2e-6, atol=1e-7)


def test_row_partition_and_slices():
    from sgracex1_amd import dist as D
    assert D.row_partition(10, 4) == [0, 2, 4, 6, 10]                 # remainder to the last (K.cpp:3522)
    rp = torch.tensor([0, 10, 10, 11, 20, 21, 40], dtype=torch.int32)
    b = D.row_partition(6, 2, rp)
    assert b[0] == 0 and b[-1] == 6 and 0 < b[1] < 6
    col = torch.arange(40, dtype=torch.int32)
    val = torch.ones(40)
    r, c, v = D.slice_rows(rp, col, val, 2, 5)
    assert r.tolist() == [0, 1, 10, 11] and c.tolist() == list(range(10, 21)) and v.numel() == 11


def test_cached_on_lives_with_the_tensor_and_follows_its_version():
    """Derived data is kept on the source tensor: rebuilt after an in-place change, never shared with
    another tensor object (not even one that reuses the storage address)."""
    import torch
    from sgracex1_amd import ops
    calls = []

    def build():
        calls.append(1)
        return len(calls)

    t = torch.arange(6)
    assert ops.cached_on(t, "k", build) == 1 and ops.cached_on(t, "k", build) == 1 and len(calls) == 1
    assert ops.cached_on(t, ("other", 3), build) == 2                  # another key on the same tensor
    t.add_(1)                                                          # version counter moves
    assert ops.cached_on(t, "k", build) == 3
    u = t.detach()                                                     # same storage, another tensor object
    assert ops.cached_on(u, "k", build) == 4 and ops.cached_on(t, "k", build) == 3


def test_quantiser_registers_round_trip_through_the_register_map():
    """init-time / per-call register writes of the reference (SG.py:334-365, :476, :1745-1838) decoded back
    into the constants the C ABI takes; absent registers mean the plain layer."""
    import numpy as np
    from sgracex1_amd import pynq_shim, quant
    from sgracex1_amd.sgrace import _program_quant_registers
    ip = pynq_shim.IP(device="cpu")
    assert ip.quant_from_registers() is None
    for bits, beta in ((8, 255), (4, 15), (2, 2), (1, 1)):
        c = quant.constants(bits)
        ip.register_map.beta_qu = beta
        _program_quant_registers(ip.register_map, c)
        q = ip.quant_from_registers()
        assert (q.w_qbits, q.scale_fea, q.internal_quantization) == (bits, c.scale_fea, c.internal_quantization)
        for name in ("w_s", "a_s", "f_s"):
            assert np.float32(1 / getattr(q, name)) == np.float32(1 / getattr(c, name))
        assert np.float32(q.deq_o) == np.float32(c.deq_o)
        s = q.as_struct(nnz_adj=10, nnz_fea=4, adj_done=True)
        assert s.qbits == bits and s.flags == 1 and s.nnz_adj == 10 and s.nnz_fea == 4
    ip.register_map.beta_qu = 7
    import pytest
    with pytest.raises(ValueError):
        ip.quant_from_registers()
    assert ip.register_map.max_fea == 0
