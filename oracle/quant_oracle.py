"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the quantised layer of the SGRACE library, the
`acc == 0`, `fake_quantization == 1` branch of FPYNQ_GAT.forward (demo/sgrace_lib/sgrace.py:563-681
with the helpers of :177-265 and the constants of :95-174, :1645-1848).  Dense N x N torch code on
the CPU, the same operators in the same order as the reference, for small cases.

Parity unpinned: the reference ships no recorded output of this branch (its notebooks print
accuracies only) and its Python module cannot be imported here (torch_geometric is absent), so this
restatement is checked by reading, not against reference outputs.
"""
import torch


def fake_quantization(x, s, z, alpha_q, beta_q, w_qbits):            # SG.py:191-235
    x_r = torch.round(1 / s * x + z, decimals=0)
    x_q = torch.clip(x_r, min=alpha_q, max=beta_q)
    return x_q / (2 ** (w_qbits - 1))


def fake_quantization_b(x, s, z):                                    # SG.py:177-182
    x_q = (1 / s * x + z)
    x_q[x_q < 0] = -0.5
    x_q[x_q >= 0] = 0.5
    return x_q


def fake_quantization_b2(x, s, z, alpha_q, beta_q):                  # SG.py:184-189
    x_r = torch.round(1 / s * x + z, decimals=0)
    return torch.clip(x_r, min=alpha_q, max=beta_q) / 2


def quantization_fbits(x, s, z, qbits):                              # SG.py:238-251 (signed)
    if qbits == 1:
        return fake_quantization_b(x, s, z)
    return fake_quantization(x, s, z, -2 ** (qbits - 1) + 1, 2 ** (qbits - 1) - 1, qbits)


def quantization_ufbits(x, s, z, qbits):                             # SG.py:253-265 (unsigned)
    if qbits == 1:
        return fake_quantization_b2(x, s, z, 0, 2 ** qbits - 1)
    return fake_quantization(x, s, z, 0, 2 ** qbits - 1, qbits)


def affine_constants(alpha, beta, alpha_q, beta_q, w_qbits):         # SG.py:95-132
    if w_qbits == 1:
        beta_o, alpha_o = beta_q / (2 ** 2), alpha_q / (2 ** 2)
    else:
        beta_o, alpha_o = beta_q / (2 ** w_qbits), alpha_q / (2 ** w_qbits)
    s_o = (beta - alpha) / (beta_o - alpha_o)
    s = (beta - alpha) / (beta_q - alpha_q)
    z = int((beta * alpha_q - alpha * beta_q) / (beta - alpha))
    return s_o, s, z


def layer(adj_dense, x, weights, attention, c, relu, compute_attention, alpha=0.2):
    """SG.py:565-667.  c: any object with w_qbits, w_s, w_z, a_s, a_z, f_s, f_z, scale_fea,
    internal_quantization, deq_o.  Returns (output, e, attentions-or-adj_q)."""
    x = x.float()
    input_q = quantization_ufbits(x, c.f_s, c.f_z, c.w_qbits)
    weights_q = quantization_fbits(weights, c.w_s, c.w_z, c.w_qbits)
    Wh = torch.mm(input_q, weights_q)
    Wh = Wh / (2 ** c.scale_fea)
    iq = c.internal_quantization
    a_min = -(2 ** iq - 1) / (2 ** iq)
    a_max = (2 ** iq - 1) / (2 ** iq)
    Wh = torch.clip(Wh, min=a_min, max=a_max)
    Wh = torch.round(Wh, decimals=(iq - 1))
    attention = quantization_fbits(attention, c.w_s, c.w_z, c.w_qbits)
    adj_d = quantization_ufbits(adj_dense.clone(), c.a_s, c.a_z, c.w_qbits)
    F = weights.shape[1]
    e = torch.matmul(Wh, attention[:F, :]) + torch.matmul(Wh, attention[F:, :]).T
    e = torch.nn.functional.leaky_relu(e, alpha)
    attention1 = torch.where(adj_d > 0, e, -9e15 * torch.ones_like(e))
    attentions = torch.nn.functional.softmax(attention1, dim=1)
    if compute_attention:
        out = torch.matmul(attentions, Wh)
    else:
        out = torch.matmul(adj_d.to_sparse(), Wh)
    if relu:
        out = torch.where(out > 0, out, 0)
    out = out * c.deq_o
    return out, e, (attentions if compute_attention else adj_d), Wh
