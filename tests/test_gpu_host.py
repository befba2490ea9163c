"""The host mirror of the reference interface on a real GPU: the notebook's register-map flow
(mmult-master.ipynb cells 16-38) through the pynq-shaped shim, the autograd Functions and modules
of molecule_gcn and of the SGRACE library against their own `acc == 0` torch twins."""
import os

import numpy as np
import pytest
import torch

from _fixtures import GOLD, half_ulp_distance, known_answers, load

pytestmark = pytest.mark.gpu


def test_notebook_register_map_flow_citeseer(oracle):
    """mmult-master.ipynb, cell by cell: allocate, load text matrices, write the registers,
    AP_START, poll AP_DONE, read D_buffer -- and compare row 0 with what the notebook recorded
    from the FPGA (cell 37) and with the oracle."""
    from sgracex1_amd import pynq_shim
    pynq_shim.install()
    from pynq import Overlay, allocate                                     # cell 2, 16
    d = load("citeseer")
    N_adj, M_fea, P_w = 3327, 3703, 16                                     # cell 6
    NNZ_adj, NNZ_fea = 12431, 105165
    ol = Overlay("gnn_all.bit")
    my_ip = ol.mmult_top_0                                                 # cell 11
    profiling_buffer = allocate(16, dtype=np.int64)
    rowPtr_fea_buffer = allocate(N_adj + 1, dtype=np.int32)
    columnIndex_fea_buffer = allocate(NNZ_fea, dtype=np.int32)
    values_fea_buffer = allocate(200000, dtype=np.float16)
    rowPtr_adj_buffer = allocate(N_adj + 1, dtype=np.int32)
    columnIndex_adj_buffer = allocate(NNZ_adj, dtype=np.int32)
    values_adj_buffer = allocate(NNZ_adj, dtype=np.float16)
    B_buffer = allocate(shape=(P_w, M_fea), dtype=np.float16)
    D_buffer = allocate(shape=(N_adj, P_w), dtype=np.float16)
    w = d["w"].astype(np.float16)                                          # cell 18: genfromtxt(dtype=float16)
    B_buffer[:] = w[:, :P_w].T
    D_buffer[:] = 1                                                        # cell 21
    rowPtr_adj_buffer[:] = d["adj_rowptr"]                                 # cell 25
    columnIndex_adj_buffer[:] = d["adj_col"]
    values_adj_buffer[:] = d["adj_val"].astype(np.float16)
    rowPtr_fea_buffer[:] = d["fea_rowptr"]                                 # cell 26
    columnIndex_fea_buffer[:] = d["fea_col"]
    values_fea_buffer[0:NNZ_fea] = d["fea_val"].astype(np.float16)
    rm = my_ip.register_map                                                # cell 31
    rm.B_offset_1 = B_buffer.physical_address
    rm.D1_offset_1 = D_buffer.physical_address
    rm.D2_offset_1 = D_buffer.physical_address + P_w * N_adj / 1
    for k in "1234":
        setattr(rm, f"rowPtr_fea{k}_offset_1", rowPtr_fea_buffer.physical_address)
        setattr(rm, f"columnIndex_fea{k}_offset_1", columnIndex_fea_buffer.physical_address)
        setattr(rm, f"values_fea{k}_offset_1", values_fea_buffer.physical_address)
        setattr(rm, f"rowPtr_adj{k}_offset_1", rowPtr_adj_buffer.physical_address)
        setattr(rm, f"columnIndex_adj{k}_offset_1", columnIndex_adj_buffer.physical_address)
        setattr(rm, f"values_adj{k}_offset_1", values_adj_buffer.physical_address)
    rm.profiling_offset_1 = profiling_buffer.physical_address
    rm.M_fea, rm.N_adj, rm.M_adj, rm.P_w = M_fea, N_adj, N_adj, P_w
    rm.relu, rm.gemm_mode, rm.bias_count = 0, 0, 0

    def run_kernel():                                                      # cell 32
        my_ip.register_map.CTRL.AP_START = 1
        kernel_done = my_ip.register_map.CTRL.AP_DONE
        while kernel_done == 0:
            kernel_done = my_ip.register_map.CTRL.AP_DONE

    run_kernel()
    result = np.zeros(shape=(N_adj, P_w), dtype=np.float16)                # cell 36
    result[:] = D_buffer[:]
    hw = np.array([float(t) for t in known_answers()["hw_row0_fp16_P16"]], dtype=np.float16)
    # cell 37 is the FPGA's half-accumulated row; the fp32-accumulated row sits inside the stated band
    np.testing.assert_allclose(result[0].astype(np.float32), hw.astype(np.float32), rtol=1e-2, atol=2e-3)
    h = lambda a: oracle.to_half(np.asarray(a, np.float32)).astype(np.float32)
    want = oracle.layer_f64(0, 0, (d["adj"][0], d["adj"][1], h(d["adj"][2])), (d["fea"][0], d["fea"][1], h(d["fea"][2])),
                            h(np.ascontiguousarray(d["w"][:, :P_w].T)), h_round=2)
    np.testing.assert_allclose(result.astype(np.float32), want, rtol=1e-2, atol=2e-3)
    # The buffers live in pinned host memory and keep device mirrors: the first start moved everything, a
    # second start with untouched buffers moves nothing up, and a second LAYER on the same graph -- new features and
    # weights written by slice assignment, as MOL cell 16 does -- uploads those and not one byte of the adjacency.
    st = my_ip.transfer_stats
    adj_bytes = rowPtr_adj_buffer.nbytes + columnIndex_adj_buffer.nbytes + values_adj_buffer.nbytes
    assert st["uploaded_bytes"] >= adj_bytes and st["reused_bytes"] == 0 and st["downloaded_bytes"] == D_buffer.nbytes
    up0 = st["uploaded_bytes"]
    run_kernel()
    assert st["uploaded_bytes"] == up0 and st["reused_bytes"] >= adj_bytes
    assert np.array_equal(np.asarray(D_buffer), result)
    values_fea_buffer[0:NNZ_fea] = d["fea_val"].astype(np.float16)         # (the same values again: a write is a write)
    B_buffer[:] = w[:, :P_w].T
    reused0 = st["reused_bytes"]
    run_kernel()
    assert st["uploaded_bytes"] - up0 == NNZ_fea * 2 + B_buffer.nbytes            # the slice that is read, not the 200000-element buffer
    assert st["reused_bytes"] - reused0 >= adj_bytes + rowPtr_fea_buffer.nbytes + columnIndex_fea_buffer.nbytes
    np.testing.assert_array_equal(np.asarray(D_buffer), result)
    # a write that goes around the buffer object (no slice assignment, no flush): the content stamp catches it
    keep = values_adj_buffer[0].copy()
    np.asarray(values_adj_buffer)[0] = 0
    run_kernel()
    changed = np.asarray(D_buffer) != result
    assert changed.any() and st["uploaded_bytes"] - up0 >= NNZ_fea * 2 + B_buffer.nbytes + values_adj_buffer.nbytes
    values_adj_buffer[0:1] = keep
    run_kernel()
    assert np.array_equal(np.asarray(D_buffer), result)
    assert not profiling_buffer[:15].any()                                 # cells 39-40: all zero
    # relu register and the dense mode through the same registers
    rm.relu = 1
    run_kernel()
    assert np.array_equal(D_buffer, np.maximum(result, np.float16(0)))
    # bias_count > 0: the kernel returns without touching D (K.cpp:3876-3889)
    D_buffer[:] = 3
    rm.bias_count = 2
    run_kernel()
    assert (D_buffer == 3).all()


def _mutag_batch(device):
    from sgracex1_amd import pyg_lite as G
    raw = np.load(os.path.join(GOLD, "mutag_raw.npz"))
    graphs = G.load_tu_raw(raw["A"], raw["graph_indicator"], raw["graph_labels"], raw["node_labels"])
    return G.collate(graphs).to(device), graphs


def test_fpynq_forward_backward_matches_torch_twin():
    """FPYNQ (forward on the kernel, fp16) against `adj @ x @ W` autograd in fp32 on the same
    symmetric graph: outputs within the fp16 band, gradients within 1e-2 relative."""
    from sgracex1_amd import molecule_gcn as M, pynq_shim
    from sgracex1_amd.pyg_lite import to_dense_adj
    dev = torch.device("cuda")
    batch, _ = _mutag_batch(dev)
    adj = to_dense_adj(batch.edge_index, batch.num_nodes)[0]
    ip = pynq_shim.Overlay("gnn_all.bit").mmult_top_0
    torch.manual_seed(0)
    layer = M.GraphConvolution_pynq(7, 64, ip).to(dev)
    x = batch.x.clone().requires_grad_(True)
    out = layer(1, 0, 1, x, adj)                                           # acc, dense, relu
    assert out.dtype == torch.float16 and out.shape == (3371, 64)
    ref_x = batch.x.clone().requires_grad_(True)
    ref = torch.relu(adj @ ref_x @ layer.weight)
    np.testing.assert_allclose(out.float().detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-2, atol=2e-3)
    assert ((out == 0) == (ref <= 0)).float().mean() > 0.999
    g = torch.randn_like(ref)
    # RPYNQ masks where the layer output is 0, exactly what relu's autograd does on the twin
    y = M.Relu_pynq()(out)
    y.backward(g.half())
    gw_dev = layer.weight.grad.clone()
    layer.weight.grad = None
    ref.backward(g)
    scale = layer.weight.grad.abs().max()
    assert (gw_dev - layer.weight.grad).abs().max() / scale < 1e-2
    assert (x.grad - ref_x.grad).abs().max() / ref_x.grad.abs().max() < 1e-2
    # dense mode (layer 2 of the notebook) and the acc == 0 twin of the module itself
    layer2 = M.GraphConvolution_pynq(64, 64, ip).to(dev)
    o2 = layer2(1, 1, 0, out.detach(), adj)
    t2 = layer2(0, 1, 0, out.detach(), adj)
    np.testing.assert_allclose(o2.detach().float().cpu().numpy(), t2.detach().cpu().numpy(), rtol=1e-2, atol=4e-3)


def test_gcn_pynq_model_acc_vs_cpu_twin():
    from sgracex1_amd import molecule_gcn as M, pynq_shim
    dev = torch.device("cuda")
    batch, _ = _mutag_batch(dev)
    ip = pynq_shim.Overlay("gnn_all.bit").mmult_top_0
    model = M.GCN_PYNQ(64, 7, 2, ip).to(dev).eval()
    with torch.no_grad():
        a = model(1, batch.x, batch.edge_index, batch.batch)
        b = model(0, batch.x, batch.edge_index, batch.batch)
    assert a.shape == (188, 2)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-2, atol=5e-3)


@pytest.mark.parametrize("compute_attention", [0, 1])
def test_gatconv_sgrace_acc_vs_dense_emulation(compute_attention):
    """GATConv_SGRACE with config.acc = 1 (kernels) against config.acc = 0 (the reference's dense
    torch emulation), forward and all three gradients, on cora."""
    from sgracex1_amd import config, sgrace
    dev = torch.device("cuda")
    d = load("cora")
    n = d["N"]
    rp, ci, _ = d["adj"]
    row = np.repeat(np.arange(n), np.diff(rp))
    keep = row != ci
    ei = torch.as_tensor(np.stack([row[keep], ci[keep]]), dtype=torch.int64, device=dev)
    edge_index, norm = sgrace.sym_norm2(ei, n, fill=1, dtype=torch.float32)
    adj = torch.sparse_coo_tensor(edge_index, norm, (n, n))
    X = torch.zeros((n, d["M_fea"]), device=dev)
    frow = np.repeat(np.arange(n), np.diff(d["fea"][0]))
    X[torch.as_tensor(frow, device=dev), torch.as_tensor(d["fea"][1].astype(np.int64), device=dev)] = 1.0
    X = X[:, :256].contiguous()                                         # keep the dense twin small
    outs = {}
    for acc in (1, 0):
        config.acc, config.compute_attention, config.fake_quantization = acc, compute_attention, 0
        config.device = "cuda"
        sgrace.init_SGRACE()
        torch.manual_seed(5)
        layer = sgrace.GATConv_SGRACE(256, 16, 1, dropout=0.1, alpha=0.2).to(dev)
        x = X.clone().requires_grad_(True)
        out = layer(compute_attention, 1, 1, x, edge_index, norm, adj)       # dense=1, relu=1
        out = sgrace.Relu_SGRACE()(out)
        torch.manual_seed(6)
        out.backward(torch.randn_like(out))
        outs[acc] = (out.detach(), x.grad.clone(), layer.weight.grad.clone(), layer.attention.grad.clone())
    config.acc = 1
    for k, (a, b) in enumerate(zip(outs[1], outs[0])):
        if k == 3 and not compute_attention:
            assert not a.any() and not b.any()
            continue
        err = (a - b).abs().max() / (b.abs().max() + 1e-12)
        assert err < 2e-3, (k, float(err))


def test_mutag_training_reaches_reference_accuracy():
    """molecule_gcn end to end (MOL cells 6-20): 188-graph batch, hidden 64, Adam lr 0.01, dropout 0.5, model seed
    12345, fp16 kernels; the notebook reports test accuracy 0.62 -> 0.76 at epoch 34 on graphs [50:100] of its
    shuffle (the permutation depends on the torch version, so 0.76 is approximate).  Both branches of the model's `acc`
    switch are trained from the same seeds on the same split: acc = 1 (the kernels) must reach the accuracy of acc = 0
    (the notebook's own torch path, the parity twin) to within one graph of the 50 (0.02), and at least 0.74."""
    from sgracex1_amd import molecule_gcn as M, pyg_lite as G, pynq_shim
    dev = torch.device("cuda")
    raw = np.load(os.path.join(GOLD, "mutag_raw.npz"))
    graphs = G.load_tu_raw(raw["A"], raw["graph_indicator"], raw["graph_labels"], raw["node_labels"])
    torch.manual_seed(12345)
    perm = torch.randperm(len(graphs)).tolist()
    graphs = [graphs[i] for i in perm]
    train, test = G.collate(graphs[:2000]).to(dev), G.collate(graphs[50:100]).to(dev)
    ip = pynq_shim.Overlay("gnn_all.bit").mmult_top_0

    def run(acc, epochs=100):
        torch.manual_seed(12345)                                   # MOL cell 18: the model's seed
        model = M.GCN_PYNQ(64, 7, 2, ip).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        crit = torch.nn.CrossEntropyLoss()
        torch.manual_seed(777)                                     # the dropout masks of both runs
        losses, accs = [], []
        for _epoch in range(epochs):
            model.train()
            opt.zero_grad()
            loss = crit(model(acc, train.x, train.edge_index, train.batch), train.y)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            model.eval()
            with torch.no_grad():
                pred = model(acc, test.x, test.edge_index, test.batch).argmax(dim=1)
            accs.append(float((pred == test.y).float().mean()))
        first = next((i + 1 for i, a in enumerate(accs) if a >= 0.76 - 1e-6), None)
        return max(accs), first, losses

    best1, first1, losses1 = run(1)
    best0, first0, losses0 = run(0)
    print(f"molecule_gcn: acc=1 best {best1:.2f} (0.76 first reached at epoch {first1}), acc=0 best {best0:.2f} "
          f"(epoch {first0}); reference notebook: 0.76 at epoch 34")
    assert losses1[-1] < losses1[0] and losses0[-1] < losses0[0]
    assert best1 >= 0.74, best1
    assert best1 >= best0 - 0.02, (best1, best0)


def test_two_layer_forward_replayed_from_a_hipgraph(oracle):
    """The launches of a forward pass captured once and replayed with new input values (sparse first
    layer, dense second, GAT third): same bits as the eager calls."""
    import numpy as np
    from _fixtures import GOLD, load
    from sgracex1_amd import graphs, ops
    from sgracex1_amd.graphed import Graphed
    d = load("cora")
    dev = torch.device("cuda")
    w2 = np.load(os.path.join(GOLD, "cora.npz"))["w2"].astype(np.float32)
    A = graphs.csr_from_numpy(*d["adj"], d["N"])
    X = graphs.csr_from_numpy(*d["fea"], d["M_fea"])
    A.plan, X.plan
    W1t = torch.as_tensor(d["Wt"], device=dev).half()
    W2t = torch.as_tensor(np.ascontiguousarray(w2.T), device=dev).half()
    att = (torch.rand(2 * W2t.shape[0], device=dev) - 0.5).half()
    D1 = torch.empty((A.n_rows, W1t.shape[0]), dtype=torch.float16, device=dev)
    D2 = torch.empty((A.n_rows, W2t.shape[0]), dtype=torch.float16, device=dev)
    D3 = torch.empty_like(D2)

    def forward():
        ops.layer_forward(A, X, W1t, relu=True, out=D1)
        ops.layer_forward(A, D1, W2t, relu=False, out=D2)
        ops.layer_forward(A, D1, W2t, relu=True, gat_attention=att, out=D3)
        return D2

    run = Graphed(forward)
    for scale in (1.0, 0.5, -2.0):
        X.val.copy_(torch.as_tensor(d["fea"][2], device=dev).half() * scale)      # new values, same buffers
        D1.zero_(), D2.zero_(), D3.zero_()
        out = run()
        torch.cuda.synchronize()
        got = (D1.clone(), out.clone(), D3.clone())
        forward()
        torch.cuda.synchronize()
        assert torch.equal(got[0], D1) and torch.equal(got[1], D2) and torch.equal(got[2], D3)
        assert D2.abs().max() > 0


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("attention,qbits,floor", [(False, 32, 0.93), (True, 32, 0.93), (False, 8, 0.9), (True, 8, 0.9)])
def test_sgrace_demo_model_trains_on_the_kernels(attention, qbits, floor):
    """The demo's two-layer model (GAT_PYNQ) trained on a planted-partition graph with the layers on the
    kernels: GCN and GAT aggregates, plain fp32 and the 8-bit quantised arithmetic."""
    import importlib.util
    from sgracex1_amd import config, sgrace
    spec = importlib.util.spec_from_file_location("sgrace_nc", os.path.join(ROOT, "examples", "sgrace_node_classification.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = config.snapshot()
    try:
        res, model, (x, ei, _y) = mod.run(attention, qbits, epochs=60, acc=1, n=2000, verbose=False)
        assert res["test_acc"] > floor, res
        # the same weights through the reference's dense emulation (acc = 0)
        model.eval()
        with torch.no_grad():
            on_gpu = model(x, ei).cpu()
        config.acc, config.device = 0, "cpu"
        sgrace.init_SGRACE()
        twin = sgrace.GAT_PYNQ(x.shape[1], 16, 1, 5)
        twin.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        twin.eval()
        with torch.no_grad():
            on_cpu = twin(x.cpu(), ei.cpu())
        close = torch.isclose(on_gpu, on_cpu, rtol=1e-3, atol=1e-3)
        assert close.float().mean() > (0.999 if qbits == 32 else 0.97), float((on_gpu - on_cpu).abs().max())
        assert (on_gpu.argmax(1) == on_cpu.argmax(1)).float().mean() > 0.98
    finally:
        config.restore(old)
        sgrace.init_SGRACE()


def test_result_buffers_are_checked_before_the_library_writes_through_them():
    from sgracex1_amd import graphs, ops
    dev = torch.device("cuda")
    A = graphs.uniform_graph(500, 3000, seed=2)
    H = torch.rand((500, 64), device=dev).half()
    ok = torch.empty((500, 64), dtype=torch.float16, device=dev)
    assert ops.spmm(A, H, out=ok) is ok
    padded = torch.zeros((500, 72), dtype=torch.float16, device=dev)
    view = padded[:, :64]
    ops.spmm(A, H, out=view)                                         # padded rows are fine for the stage ...
    assert torch.equal(view, ok) and not padded[:, 64:].any()
    for bad in (torch.empty((500, 32), dtype=torch.float16, device=dev), torch.empty((499, 64), dtype=torch.float16, device=dev),
                torch.empty((500, 64), dtype=torch.float32, device=dev), torch.empty((500, 64), dtype=torch.float16),
                torch.empty((64, 500), dtype=torch.float16, device=dev).t()):
        with pytest.raises(ValueError):
            ops.spmm(A, H, out=bad)
    Wt = torch.rand((64, 64), device=dev).half()
    with pytest.raises(ValueError):
        ops.layer_forward(A, H, Wt, out=view)                        # ... the layer writes D densely
    with pytest.raises(ValueError):
        ops.gat_aggregate(A, H, torch.rand(128, device=dev).half(), out=torch.empty((500, 65), dtype=torch.float16, device=dev))


def test_stream_copy_moves_every_byte():
    """sgx_stream_copy (the kernel bench.py measures the attainable HBM rate with): sizes around the 4-deep
    unrolled stride, bit-equal; misaligned and ragged requests are refused."""
    import ctypes
    from sgracex1_amd._lib import lib
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n16 in (1, 255, 256 * 16 * 256 * 4 - 1, 256 * 16 * 256 * 4 + 77, 3_000_001):
        src = torch.randint(-2**31, 2**31 - 1, (n16 * 4,), dtype=torch.int32, device="cuda")
        dst = torch.zeros_like(src)
        assert lib.sgx_stream_copy(dst.data_ptr(), src.data_ptr(), n16 * 16, stream) == 0
        assert torch.equal(dst, src)
    assert lib.sgx_stream_copy(dst.data_ptr(), src.data_ptr(), 0, stream) == 0
    assert lib.sgx_stream_copy(dst.data_ptr(), src.data_ptr(), 24, stream) != 0
    assert lib.sgx_stream_copy(dst.data_ptr() + 4, src.data_ptr(), 32, stream) != 0


def test_config_layer_order_auto_reaches_the_modules():
    """config.layer_order = 'auto': a dense layer narrower at its input than at its output (7 -> 64 on the MUTAG
    batch, dense mode) aggregates first; same result within the layer's band, the flag back to 'reference'
    restores the reference's dataflow bit for bit."""
    import sgracex1_amd.config as config
    from sgracex1_amd import molecule_gcn as M, pynq_shim
    from sgracex1_amd.pyg_lite import to_dense_adj
    dev = torch.device("cuda")
    batch, _ = _mutag_batch(dev)
    adj = to_dense_adj(batch.edge_index, batch.num_nodes)[0]
    ip = pynq_shim.Overlay("gnn_all.bit").mmult_top_0
    torch.manual_seed(3)
    layer = M.GraphConvolution_pynq(7, 64, ip).to(dev)
    x = (batch.x + 0.25 * torch.rand_like(batch.x)).detach()                  # dense mode wants a dense X
    assert config.layer_order == "reference"
    base = layer(1, 1, 1, x, adj).detach()
    try:
        config.layer_order = "auto"
        swapped = layer(1, 1, 1, x, adj).detach()
        config.layer_order = "columns_first"
        with pytest.raises(ValueError):
            layer(1, 1, 1, x, adj)
    finally:
        config.layer_order = "reference"
    assert torch.equal(layer(1, 1, 1, x, adj).detach(), base)
    assert not torch.equal(swapped, base)                                     # another association: other roundings
    np.testing.assert_allclose(swapped.float().cpu().numpy(), base.float().cpu().numpy(), rtol=1e-2, atol=2e-3)


def test_shim_verify_mode_catches_what_the_content_stamp_cannot(monkeypatch):
    """A write that goes around the buffer object AND misses the bytes the content stamp samples (large buffers are
    fingerprinted by their ends and a strided sample) leaves a stale device mirror: SGX_SHIM_VERIFY=1 compares every mirror
    in full and says so; flush() -- pynq's own call for "the host wrote" -- repairs it."""
    from sgracex1_amd import pynq_shim
    n, m, p = 4096, 256, 16
    rng = np.random.default_rng(3)
    ip = pynq_shim.Overlay("gnn_all.bit").mmult_top_0
    rm = ip.register_map
    deg = rng.integers(1, 6, n)
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    nnz = int(rp[-1])
    bufs = {k: pynq_shim.allocate(shape, dt) for k, shape, dt in [
        ("rp", n + 1, np.int32), ("ci", nnz, np.int32), ("va", nnz, np.float16), ("X", n * m, np.float16),
        ("B", p * m, np.float16), ("D", n * p, np.float16)]}
    bufs["rp"][:] = rp
    bufs["ci"][:] = rng.integers(0, n, nnz)
    bufs["va"][:] = rng.random(nnz).astype(np.float16)
    bufs["X"][:] = rng.random(n * m).astype(np.float16)
    bufs["B"][:] = (rng.random(p * m) - 0.5).astype(np.float16)
    rm.N_adj = rm.M_adj = n
    rm.M_fea, rm.P_w, rm.relu, rm.gemm_mode = m, p, 0, 1
    rm.rowPtr_adj1_offset_1, rm.columnIndex_adj1_offset_1 = bufs["rp"].physical_address, bufs["ci"].physical_address
    rm.values_adj1_offset_1, rm.values_fea1_offset_1 = bufs["va"].physical_address, bufs["X"].physical_address
    rm.B_offset_1, rm.D1_offset_1 = bufs["B"].physical_address, bufs["D"].physical_address
    rm.CTRL.AP_START = 1
    first = np.asarray(bufs["D"]).copy()
    assert bufs["X"].nbytes > (1 << 20)
    raw = np.asarray(bufs["X"])
    raw[(4096 + 200) // 2] += np.float16(1)                  # around the buffer object, between two sampled blocks
    rm.CTRL.AP_START = 1
    assert np.array_equal(np.asarray(bufs["D"]), first)       # the stale mirror was used: the documented limit of the stamp
    monkeypatch.setenv("SGX_SHIM_VERIFY", "1")
    with pytest.raises(RuntimeError, match="stale"):
        rm.CTRL.AP_START = 1
    bufs["X"].flush()
    rm.CTRL.AP_START = 1
    assert not np.array_equal(np.asarray(bufs["D"]), first)
