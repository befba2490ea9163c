"""The quantised layer of the SGRACE library on the CPU side: the constants `init_SGRACE` derives
(SG.py:95-174, :1645-1848) against hand-computed values, and the package's `acc == 0` dense twin
against the oracle restatement of SG.py:565-667.  Parity unpinned: the reference holds no recorded
output of this branch (see oracle/quant_oracle.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import quant_oracle as QO


def test_constants_match_hand_computed_values():
    from sgracex1_amd import quant
    c = quant.constants(8)
    # signed weights in [-1, 1] on -127..127; unsigned adjacency/features in [0, 1] on 0..255
    assert c.w_s == 2 / 254 and c.w_z == 0 and c.a_s == 1 / 255 and c.f_s == 1 / 255 and c.a_z == 0 and c.f_z == 0
    w_s_o = 2 / (254 / 256)
    u_s_o = 1 / (255 / 256)
    assert math.isclose(c.deq_o, w_s_o * u_s_o * u_s_o * 2, rel_tol=1e-15)
    assert (c.scale_fea, c.internal_quantization) == (4, 16)
    c4 = quant.constants(4)
    assert c4.w_s == 2 / 14 and c4.f_s == 1 / 15 and (c4.scale_fea, c4.internal_quantization) == (3, 8)
    assert math.isclose(c4.deq_o, (2 / (14 / 16)) * (1 / (15 / 16)) ** 2 * 2, rel_tol=1e-15)
    c2 = quant.constants(2)
    assert math.isclose(c2.w_s, 0.2 / 2) and math.isclose(c2.a_s, 0.1 / 3) and c2.f_s == 1 / 3
    assert (c2.scale_fea, c2.internal_quantization) == (3, 4)
    c1 = quant.constants(1)
    # one bit: the output grid is divided by 2^2 (SG.py:110-112)
    assert math.isclose(c1.w_s, 0.2 / 2) and math.isclose(c1.deq_o, (0.2 / 0.5) * (1 / 0.25) * (0.1 / 0.25) * 2)
    assert (c1.scale_fea, c1.internal_quantization) == (2, 4)
    for bits in (8, 4, 2, 1):
        c = quant.constants(bits)
        for name, (alpha, beta, signed) in {"w": (-1.0 if bits > 2 else -0.1, 1.0 if bits > 2 else 0.1, True),
                                            "a": (0.0, 1.0 if bits > 2 else 0.1, False)}.items():
            if signed:
                aq, bq = (-1, 1) if bits == 1 else (-2 ** (bits - 1) + 1, 2 ** (bits - 1) - 1)
            else:
                aq, bq = 0, 2 ** bits - 1
            _s_o, s, z = QO.affine_constants(alpha, beta, aq, bq, bits)
            assert getattr(c, name + "_s") == s and getattr(c, name + "_z") == z
        assert c.second_layer().deq_o == c.deq_o2 and c.deq_o2 == c.deq_o     # equal ranges in the live tables
    with pytest.raises(ValueError):
        quant.constants(3)
    assert quant.constants(8, scale_fea=6).scale_fea == 6


def _case(n, m, p, seed):
    g = torch.Generator().manual_seed(seed)
    adj = (torch.rand((n, n), generator=g) < 0.15).float()
    adj = ((adj + adj.t() + torch.eye(n)) > 0).float()
    deg = adj.sum(1)
    adj = adj / torch.sqrt(deg[:, None] * deg[None, :])
    x = torch.rand((n, m), generator=g) * (torch.rand((n, m), generator=g) < 0.4)
    w = (torch.rand((m, p), generator=g) * 2 - 1) * 0.6
    att = (torch.rand((2 * p, 1), generator=g) * 2 - 1) * 0.6
    return adj, x, w, att


@pytest.mark.parametrize("bits", [8, 4, 2, 1])
@pytest.mark.parametrize("gat", [0, 1])
def test_dense_twin_equals_the_restatement(bits, gat):
    from sgracex1_amd import config, quant, sgrace
    adj, x, w, att = _case(40, 23, 8, 5 + bits)
    c = quant.constants(bits)
    want, _e, _p, _wh = QO.layer(adj, x, w, att, c, relu=1, compute_attention=gat)
    old = config.snapshot()
    try:
        config.acc, config.fake_quantization, config.w_qbits, config.compute_attention = 0, 1, bits, gat
        config.float_type = np.float32
        sgrace.init_SGRACE()
        layer = sgrace.GATConv_SGRACE(23, 8)
        with torch.no_grad():
            layer.weight.copy_(w)
            layer.attention.copy_(att)
        idx = adj.nonzero().t()
        got = layer(gat, 1, 1, x, idx, adj[idx[0], idx[1]], adj.to_sparse())
        assert torch.equal(got, want)
        assert got.abs().max() > 0
        # straight-through backward: gradients flow to the unquantised operands
        got.sum().backward()
        assert layer.weight.grad is not None and torch.isfinite(layer.weight.grad).all()
    finally:
        config.restore(old)
        sgrace.init_SGRACE()


def test_quantisers_on_known_points():
    c = type("C", (), {})()
    x = torch.tensor([0.0, 0.2, 0.5, 0.5019608, 1.0, 1.7, -0.3])
    # 8 bit unsigned, s = 1/255: 0.5 * 255 = 127.5 -> 128 (half to even), clip at 255, / 128
    q = QO.quantization_ufbits(x, 1 / 255, 0, 8)
    assert q.tolist() == [0.0, 51 / 128, 128 / 128, 128 / 128, 255 / 128, 255 / 128, 0.0]
    w = torch.tensor([-1.0, -0.004, 0.0, 0.3, 2.0])
    qs = QO.quantization_fbits(w, 2 / 254, 0, 8)
    assert qs.tolist() == [-127 / 128, -1 / 128, 0.0, 38 / 128, 127 / 128]
    assert QO.quantization_fbits(w.clone(), 0.1, 0, 1).tolist() == [-0.5, -0.5, 0.5, 0.5, 0.5]
    assert QO.quantization_ufbits(torch.tensor([0.0, 0.04, 0.06, 0.3]), 0.1, 0, 1).tolist() == [0.0, 0.0, 0.5, 0.5]
