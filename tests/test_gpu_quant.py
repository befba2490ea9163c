"""The quantised layer on the GPU (sgx_fake_quantize, sgx_requantize, sgx_layer_forward with a
sgx_quant block) against the CPU restatement of SG.py:565-667 (oracle/quant_oracle.py).  The two
rounding kernels are compared bit for bit; the layer within the fp32 band of the plain fp32 layer
(sums are formed in another order than torch.mm / torch.matmul).  Parity unpinned -- see the oracle."""
import numpy as np
import pytest
import torch

from oracle import quant_oracle as QO

pytestmark = pytest.mark.gpu
dev = torch.device("cuda")


def _points(scale, n, seed):
    """random values plus every half-way point of the grid (round-half-even) and out-of-range ones"""
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, generator=g) * 2.6 - 1.3) * scale * 300
    k = torch.arange(-300, 300, dtype=torch.float64)
    ties = ((k + 0.5) * scale).float()
    return torch.cat([x, ties, torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1e30, -1e30])])


@pytest.mark.parametrize("bits", [8, 4, 2, 1])
@pytest.mark.parametrize("signed", [0, 1])
def test_fake_quantize_bit_exact(bits, signed):
    from sgracex1_amd import ops, quant
    c = quant.constants(bits)
    s, z = (c.w_s, c.w_z) if signed else (c.f_s, c.f_z)
    x = _points(s, 50_000, bits * 2 + signed)
    want = (QO.quantization_fbits if signed else QO.quantization_ufbits)(x.clone(), s, z, bits)
    got = ops.fake_quantize(x.to(dev), signed, bits, s, z)
    assert torch.equal(got.cpu(), want)
    assert got.unique().numel() <= (2 ** bits if not signed else max(2, 2 ** bits - 1))


@pytest.mark.parametrize("iq,scale_fea", [(16, 4), (8, 3), (4, 3), (4, 2), (16, 0)])
def test_requantize_bit_exact(iq, scale_fea):
    from sgracex1_amd import ops
    g = torch.Generator().manual_seed(iq + scale_fea)
    H = (torch.rand((3001, 37), generator=g) * 2 - 1) * 2 ** scale_fea * 1.2
    H[0, :8] = torch.tensor([0.0, -0.0, 1e-9, 0.5, -0.5, 2.0 ** scale_fea, -2.0 ** scale_fea, 0.0625])
    want = H / (2 ** scale_fea)
    want = torch.clip(want, min=-(2 ** iq - 1) / (2 ** iq), max=(2 ** iq - 1) / (2 ** iq))
    want = torch.round(want, decimals=iq - 1)
    buf = torch.zeros((3001, 40), device=dev)                       # padded leading dimension, pad must stay 0
    buf[:, :37] = H.to(dev)
    ops.requantize_(buf[:, :37], scale_fea, iq)
    assert torch.equal(buf[:, :37].cpu(), want)
    assert (buf[:, 37:] == 0).all()


def _graph_case(n, m, p, seed, density=0.02):
    g = torch.Generator().manual_seed(seed)
    adj = (torch.rand((n, n), generator=g) < density).float()
    adj = ((adj + adj.t() + torch.eye(n)) > 0).float()
    deg = adj.sum(1)
    adj = adj / torch.sqrt(deg[:, None] * deg[None, :])
    x = torch.rand((n, m), generator=g) * (torch.rand((n, m), generator=g) < 0.3)
    w = (torch.rand((m, p), generator=g) * 2 - 1) * 0.7
    att = (torch.rand((2 * p, 1), generator=g) * 2 - 1) * 0.7
    return adj, x, w, att


@pytest.mark.parametrize("bits", [8, 4, 2, 1])
@pytest.mark.parametrize("gat", [0, 1])
@pytest.mark.parametrize("sparse_x", [0, 1])
def test_quantised_layer_matches_restatement(bits, gat, sparse_x):
    from sgracex1_amd import ops, quant
    n, m, p = 700, 150, 24
    adj, x, w, att = _graph_case(n, m, p, 31 + bits + gat)
    c = quant.constants(bits)
    want, e_d, p_d, _wh = QO.layer(adj, x, w, att, c, relu=1, compute_attention=gat)
    A = ops.Csr.from_dense(adj.to(dev), torch.float32)
    X = ops.Csr.from_dense(x.to(dev), torch.float32) if sparse_x else x.to(dev)
    Wt = w.t().contiguous().to(dev)
    res = ops.layer_forward(A, X, Wt, relu=True, gat_attention=att.reshape(-1).to(dev) if gat else None, quant=c,
                            want_edge_outputs=bool(gat))
    got = (res[0] if gat else res).cpu()
    tol = dict(rtol=2e-5, atol=2e-6 * c.deq_o)
    assert torch.allclose(got, want, **tol), float((got - want).abs().max())
    assert want.abs().max() > 0
    if gat:
        idx = adj.nonzero().t()
        kept = p_d[idx[0], idx[1]]                                  # softmax of the restatement on the stored edges
        live = QO.quantization_ufbits(adj[idx[0], idx[1]].clone(), c.a_s, c.a_z, bits) > 0
        S = res[2].cpu()
        # edges the quantiser zeroed are masked (SG.py:640): 0, or 1/N in rows left without any live edge
        assert torch.allclose(S, kept, rtol=1e-4, atol=1e-6)
        assert bool(live.all()) or (S[~live] <= 1.0 / n + 1e-9).all()
    # the adjacency quantised once by the host and flagged as such gives the same bits
    Aq = ops.Csr(A.rowptr, A.col, ops.fake_quantize(A.val, 0, bits, c.a_s, c.a_z), A.n_cols)
    res2 = ops.layer_forward(Aq, X, Wt, relu=True, gat_attention=att.reshape(-1).to(dev) if gat else None, quant=c,
                             adj_quantized=True)
    assert torch.equal(res2.cpu(), got)
    if not gat:                                                     # ... and so does quantising it inside the call
        res3 = ops.layer_forward(A, X, Wt, relu=True, quant=c, cache_quantized_adj=False)
        assert torch.equal(res3.cpu(), got)


def test_quantised_layer_rejects_half_tensors():
    from sgracex1_amd import ops, quant
    adj, x, w, _ = _graph_case(50, 10, 8, 3)
    A = ops.Csr.from_dense(adj.to(dev), torch.float16)
    with pytest.raises(TypeError):
        ops.layer_forward(A, x.to(dev).half(), w.t().contiguous().to(dev).half(), quant=quant.constants(8))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_gat_row_without_live_edge_follows_the_dense_emulation(dtype):
    """SG.py:638-641: a row whose every entry is masked is constant before the softmax, so the
    softmax is uniform over all N nodes and the row receives the mean of all rows of Wh."""
    from sgracex1_amd import ops
    n, p = 300, 24
    adj, _x, _w, att = _graph_case(n, 10, p, 13, density=0.03)
    adj[7, :] = 0                                   # no entry at all
    adj[19, :] = -adj[19, :]                        # entries, none positive
    g = torch.Generator().manual_seed(2)
    Wh = (torch.rand((n, p), generator=g) - 0.3).to(dtype).float()
    e = torch.nn.functional.leaky_relu(Wh @ att[:p] + (Wh @ att[p:]).T, 0.2)
    attn = torch.softmax(torch.where(adj > 0, e, -9e15 * torch.ones_like(e)), dim=1)
    want = attn @ Wh
    A = ops.Csr.from_dense(adj.to(dev), dtype)
    assert A.has_dead_rows
    got, E, S = ops.gat_aggregate(A, Wh.to(dev).to(dtype), att.reshape(-1).to(dev).to(dtype), want_edge_outputs=True)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-3)
    assert torch.allclose(got.float().cpu(), want, **tol)
    assert torch.allclose(got[7].float().cpu(), Wh.mean(0), **tol) and torch.allclose(got[19].float().cpu(), Wh.mean(0), **tol)
    rp = A.rowptr.cpu()
    assert torch.allclose(S[rp[19]:rp[20]].cpu(), torch.full((int(rp[20] - rp[19]),), 1.0 / n))
    # the layer entry point sets the same flag from the adjacency it is given
    adj2 = adj.clone()
    adj2[19, :] = 0
    A2 = ops.Csr.from_dense(adj2.to(dev), dtype)
    X = torch.eye(n, dtype=dtype, device=dev)
    out = ops.layer_forward(A2, X, Wh.t().contiguous().to(dev).to(dtype), gat_attention=att.reshape(-1).to(dev).to(dtype))
    assert torch.allclose(out.float().cpu(), want, **tol)


@pytest.mark.parametrize("bits,gat", [(8, 0), (8, 1), (4, 0), (1, 1)])
def test_sgrace_layers_quantised_on_the_gpu_match_the_dense_twin(bits, gat):
    """GATConv_SGRACE with config.acc = 1 (kernels) against config.acc = 0 (dense emulation), two
    stacked layers so that the hardware path's alternating constant sets are exercised."""
    from sgracex1_amd import config, sgrace
    adj, x, w1, att1 = _graph_case(300, 60, 16, 77 + bits)
    _, _, w2, att2 = _graph_case(300, 16, 16, 78 + bits)
    idx = adj.nonzero().t()
    norm = adj[idx[0], idx[1]]
    old = config.snapshot()
    outs = {}
    try:
        for acc in (0, 1):
            config.acc, config.fake_quantization, config.w_qbits, config.compute_attention = acc, 1, bits, gat
            config.float_type = np.float32
            sgrace.init_SGRACE()
            d = dev if acc else torch.device("cpu")
            l1, l2 = sgrace.GATConv_SGRACE(60, 16).to(d), sgrace.GATConv_SGRACE(16, 16).to(d)
            with torch.no_grad():
                l1.weight.copy_(w1), l1.attention.copy_(att1), l2.weight.copy_(w2[:16]), l2.attention.copy_(att2)
            h = l1(gat, 1, 1, x.to(d), idx.to(d), norm.to(d), adj.to_sparse().to(d) if not acc else adj.to(d).to_sparse())
            # keep the second layer's input inside the feature range [0, 1] of the tables
            h = torch.clamp(h, 0, 1)
            out = l2(gat, 1, 0, h, idx.to(d), norm.to(d), adj.to_sparse().to(d) if not acc else adj.to(d).to_sparse())
            out.sum().backward()
            outs[acc] = (out.detach().cpu(), l1.weight.grad.detach().cpu())
        # a 1-ulp difference in layer 1 can move a layer-2 input across a rounding boundary of the quantiser
        close = torch.isclose(outs[1][0], outs[0][0], rtol=1e-4, atol=1e-5)
        assert close.float().mean() > 0.99, float((outs[1][0] - outs[0][0]).abs().max())
        assert bits == 1 or outs[0][0].abs().max() > 0     # one bit: layer-2 inputs below 0.5 quantise to 0
    finally:
        config.restore(old)
        sgrace.init_SGRACE()


def test_register_path_carries_the_quantiser_registers():
    """The notebook flow: host buffers, registers programmed as SG.py:334-365 / :476, AP_START."""
    from sgracex1_amd import pynq_shim, quant
    from sgracex1_amd.sgrace import _program_quant_registers
    n, m, p = 200, 40, 8
    adj, x, w, att = _graph_case(n, m, p, 91, density=0.05)
    c = quant.constants(8)
    want, _e, _p, _wh = QO.layer(adj, x, w, att, c, relu=1, compute_attention=0)
    ol = pynq_shim.Overlay("gat_all_unsigned.bit")
    ip = ol.mmult_top_0
    rm = ip.register_map
    idx = adj.nonzero().t().numpy()
    nnz = idx.shape[1]
    bufs = {name: pynq_shim.allocate(shape, dt) for name, shape, dt in [
        ("row", nnz, np.int32), ("col", nnz, np.int32), ("val", nnz, np.float32), ("B", p * m, np.float32),
        ("X", n * m, np.float32), ("D", n * p, np.float32)]}
    bufs["row"][:], bufs["col"][:] = idx[0], idx[1]
    bufs["val"][:] = adj.numpy()[idx[0], idx[1]]
    bufs["B"][:] = w.t().contiguous().numpy().reshape(-1)
    bufs["X"][:] = x.numpy().reshape(-1)
    rm.gemm_mode, rm.relu, rm.gat_mode = 1, 1, 0
    rm.N_adj, rm.M_adj, rm.M_fea, rm.P_w, rm.nnz_adj1 = n, n, m, p, nnz
    rm.rowPtr_adj1_offset_1, rm.columnIndex_adj1_offset_1 = bufs["row"].physical_address, bufs["col"].physical_address
    rm.values_adj1_offset_1, rm.B_offset_1 = bufs["val"].physical_address, bufs["B"].physical_address
    rm.values_fea1_offset_1, rm.D1_offset_1 = bufs["X"].physical_address, bufs["D"].physical_address
    rm.beta_qu, rm.f_align = 255, 0
    _program_quant_registers(rm, c)
    qc = ip.quant_from_registers()
    assert qc.w_qbits == 8 and qc.scale_fea == 4 and qc.internal_quantization == 16
    assert np.float32(1 / qc.f_s) == np.float32(1 / c.f_s) and np.float32(qc.deq_o) == np.float32(c.deq_o)
    rm.CTRL.AP_START = 1
    assert rm.CTRL.AP_DONE == 1
    got = torch.from_numpy(np.asarray(bufs["D"]).reshape(n, p).copy())
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-6 * c.deq_o)


# ---- integer operands on the int8 matrix cores (csrc/quant_i8.hip, SGX_QUANT_INT8) -------------------------------
@pytest.mark.parametrize("bits", [8, 4, 2, 1])
@pytest.mark.parametrize("signed", [0, 1])
def test_integer_codes_are_the_grid_points(bits, signed):
    """sgx_quantize_codes_i8: code / 2^(bits-1) (1 bit: code / 2) is exactly what the fp32 quantiser returns, at every
    tie point of the grid and outside its range; unsigned 8-bit codes are stored minus 128; pad columns are 0."""
    from sgracex1_amd import ops, quant
    c = quant.constants(bits)
    s, z = (c.w_s, c.w_z) if signed else (c.f_s, c.f_z)
    x = _points(s, 50_000 - 50_606 % 37, bits * 2 + signed)
    x = x[: x.numel() // 37 * 37].reshape(-1, 37).contiguous().to(dev)
    codes, bias = ops.quantize_codes_i8(x, signed, bits, s, z)
    assert bias == (128 if (bits == 8 and not signed) else 0) and codes.shape[1] == 48 and not codes[:, 37:].any()
    frac = 1 if bits == 1 else bits - 1
    value = (codes[:, :37].to(torch.int32) + bias).float() / 2 ** frac
    assert torch.equal(value, ops.fake_quantize(x, signed, bits, s, z))


@pytest.mark.parametrize("bits,n,M,P", [(8, 5000, 100, 47), (8, 3001, 128, 256), (8, 777, 602, 128), (8, 40, 7, 3),
                                        (4, 5000, 300, 64), (2, 1000, 130, 24), (1, 1000, 65, 16)])
def test_int8_matrix_cores_sum_exactly(bits, n, M, P):
    """sgx_xw_dense_i8 without its epilogue against the integer product of the codes formed in int64: the same number
    in every element (MFMA operand layout, the -128 repair of unsigned 8-bit codes, ragged K and P, rows past the last
    tile)."""
    from sgracex1_amd import ops, quant
    c = quant.constants(bits)
    g = torch.Generator(device=dev)
    g.manual_seed(bits * 1000 + M)
    x = torch.rand((n, M), generator=g, device=dev) * 1.2 - 0.1
    w = (torch.rand((P, M), generator=g, device=dev) * 2 - 1) * 1.1 * (c.w_s * 2 ** (bits - 1))
    Xc, xb = ops.quantize_codes_i8(x, 0, bits, c.f_s, c.f_z)
    Wc, wb = ops.quantize_codes_i8(w, 1, bits, c.w_s, c.w_z)
    assert wb == 0
    H = ops.xw_dense_i8(Xc, Wc, M, bits)
    exact = ((Xc[:, :M].double() + xb) @ Wc[:, :M].double().t())          # integers below 2^53: exact in double
    frac = 1 if bits == 1 else bits - 1
    assert exact.abs().max() < 2 ** 24 or bits == 8                  # (602 x 255 x 127 passes 2^24: rounded once, below)
    want = (exact / 4.0 ** frac).float()
    assert torch.equal(H, want)
    assert H.abs().max() > 0


@pytest.mark.parametrize("bits,gat,m", [(8, 0, 150), (8, 1, 150), (4, 0, 150), (4, 1, 300), (2, 0, 150), (1, 1, 150), (8, 0, 602)])
def test_quantised_layer_with_integer_operands(bits, gat, m):
    """The layer with SGX_QUANT_INT8 against its fp32 form: the same bits wherever the fp32 form's sums are exact
    (products of codes sum below 2^24), fp32 rounding of H apart otherwise (M_fea 602 at 8 bits) -- and inside the
    same band of the CPU restatement."""
    from sgracex1_amd import ops, quant
    n, p = 700, 24
    adj, x, w, att = _graph_case(n, m, p, 77 + bits + gat)
    c = quant.constants(bits)
    A = ops.Csr.from_dense(adj.to(dev), torch.float32)
    X, Wt = x.to(dev), w.t().contiguous().to(dev)
    kw = dict(relu=True, gat_attention=att.reshape(-1).to(dev) if gat else None, quant=c)
    ref = ops.layer_forward(A, X, Wt, **kw)
    got = ops.layer_forward(A, X, Wt, quant_int8=True, **kw)
    if m <= 500:
        assert torch.equal(got, ref)
    else:
        assert torch.allclose(got, ref, rtol=2e-5, atol=2e-6 * c.deq_o)
    want, _e, _p, _wh = QO.layer(adj, x, w, att, c, relu=1, compute_attention=gat)
    assert torch.allclose(got.cpu(), want, rtol=2e-5, atol=2e-6 * c.deq_o)
    # a sparse X keeps the fp32 form (the flag is ignored where it does not apply) -- which IS the integer result while a
    # row's products of codes sum below 2^24: every term and partial sum is a multiple of 2^-2(b-1) that fp32 holds exactly,
    # whatever the order of the sparse kernel's fma chain.  The same X stored densely through the int8 matrix cores says so.
    Xs = ops.Csr.from_dense(X, torch.float32)
    sparse = ops.layer_forward(A, Xs, Wt, **kw)
    assert torch.equal(ops.layer_forward(A, Xs, Wt, quant_int8=True, **kw), sparse)
    if m <= 500:
        assert torch.equal(sparse, got)


def _workspace_bytes(M_fea, P, flags):
    """sgx_layer_workspace_bytes of a dense-feature quantised layer: the integer form carves byte code matrices
    (M_fea rounded up to 16 per row) where the fp32 form carves float copies -- which form the library chose shows there."""
    import ctypes
    from sgracex1_amd import _lib, quant
    d = _lib.LayerDesc()
    d.gemm_mode, d.relu, d.gat_mode = 1, 1, 0
    d.N_adj = d.M_adj = 1000
    d.M_fea, d.P_w, d.dtype, d.acc_mode, d.spmm_block = M_fea, P, _lib.SGX_F32, _lib.SGX_ACC_F32, 1
    qs = quant.constants(8).as_struct(nnz_adj=5000, adj_done=True)
    qs.flags |= flags
    d.quant = ctypes.pointer(qs)
    return _lib.lib.sgx_layer_workspace_bytes(ctypes.byref(d))


def test_integer_operands_are_chosen_by_shape():
    """SGX_QUANT_INT8_AUTO: the integer form exactly where the dense features are wider than 128 columns (the fp32
    product's weights-stationary kernels end there); a forced flag and no flag at all are the two forms themselves."""
    from sgracex1_amd import _lib
    for M in (64, 128):
        assert _workspace_bytes(M, 64, _lib.SGX_QUANT_INT8_AUTO) == _workspace_bytes(M, 64, 0) != _workspace_bytes(M, 64, _lib.SGX_QUANT_INT8)
    for M in (129, 200, 602):
        assert _workspace_bytes(M, 64, _lib.SGX_QUANT_INT8_AUTO) == _workspace_bytes(M, 64, _lib.SGX_QUANT_INT8) != _workspace_bytes(M, 64, 0)
    # outputs wider than the integer kernel takes (P_w > 256) keep the fp32 form under either flag
    assert _workspace_bytes(602, 300, _lib.SGX_QUANT_INT8_AUTO) == _workspace_bytes(602, 300, 0) == _workspace_bytes(602, 300, _lib.SGX_QUANT_INT8)


@pytest.mark.parametrize("gat", [0, 1])
def test_hardware_quantize_reaches_the_int8_matrix_cores_through_the_library_layer(gat, monkeypatch):
    """config.hardware_quantize = 1; init_SGRACE(); GATConv_SGRACE(...): a dense-feature layer wider than 128 columns
    runs its X.W on the int8 matrix cores (the flag FPYNQ_GAT hands to the layer is recorded), and -- 200 columns of
    8-bit codes sum below 2^24 -- gives the bits of the fp32 emulation (config.fake_quantization alone), forward and
    backward."""
    from sgracex1_amd import config, ops, sgrace
    n, m, p = 600, 200, 32
    adj, x, w, att = _graph_case(n, m, p, 123 + gat)
    idx = adj.nonzero().t()
    norm = adj[idx[0], idx[1]]
    seen = []
    real = ops.layer_forward

    def spy(*a, **kw):
        seen.append((kw.get("quant_int8"), a[1].shape[1] if isinstance(a[1], torch.Tensor) else None))
        return real(*a, **kw)

    monkeypatch.setattr(ops, "layer_forward", spy)
    old = config.snapshot()
    outs = {}
    try:
        for hw in (0, 1):
            config.acc, config.fake_quantization, config.hardware_quantize, config.w_qbits = 1, 1, hw, 8
            config.compute_attention, config.float_type = gat, np.float32
            sgrace.init_SGRACE()
            layer = sgrace.GATConv_SGRACE(m, p).to(dev)
            with torch.no_grad():
                layer.weight.copy_(w), layer.attention.copy_(att)
            seen.clear()
            out = layer(gat, 1, 1, x.to(dev), idx.to(dev), norm.to(dev), adj.to(dev).to_sparse())      # dense = 1
            out.sum().backward()
            assert seen == [("auto" if hw else False, m)]
            outs[hw] = (out.detach().clone(), layer.weight.grad.detach().clone())
        assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
        assert outs[1][0].abs().max() > 0
    finally:
        config.restore(old)
        sgrace.init_SGRACE()
