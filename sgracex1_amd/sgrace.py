"""Host side of demo/sgrace_lib/sgrace.py (SG.py), same public names and signatures:

    sym_norm2                 SG.py:18-51
    RPYNQ                     SG.py:267-296
    FPYNQ_GAT                 SG.py:298-1126   forward(ctx, my_ip, self, adj, nnz_adj, input, weights,
                                               attention, out_features, dropout, relu)
    Relu_SGRACE               SG.py:1142-1162
    GATConv_SGRACE            SG.py:1164-1260  forward(compute_attention, dense, relu, input,
                                               edge_index, norm, adj)
    GAT_PYNQ                  demo/emulation/demo_sgrace.py:271-400 (the demo's two-layer model)
    init_SGRACE               SG.py:1271

`config.acc == 1` runs the layer on the GPU through the C ABI (GCN aggregate or single-head GAT
edge softmax, selected by `compute_attention` -> register gat_mode); `config.acc == 0` is the
reference's dense torch emulation (SG.py:563-681) kept as the parity twin.

Quantised bitstream (SG.py:53-265, :570-667, :1645-1848): with `config.w_qbits` in {8, 4, 2, 1} and
`config.fake_quantization == 1` (or `config.hardware_quantize == 1`), `init_SGRACE` derives the
constants (quant.py) and the layer runs with the quantised arithmetic -- on the GPU kernels for
`acc == 1` (the registers scale_fea, deq_factor, quantization_scale_*, quantized_multiplier are
programmed as the reference does, alternating between the layer-1 and layer-2 sets), in the dense
emulation for `acc == 0`.  The backward pass uses the unquantised operands, as in the reference.

Backward mirrors SG.py:884-1126 (the `accb == 0` branch) on the edge list instead of dense
N x N matrices: grad_input = P @ (g @ W^T), grad_weights = X^T @ (P @ g) with P = the attention
matrix (GAT) or adj (GCN) -- like the reference, P and not P^T -- and for GAT the attention-vector
gradient through softmax and LeakyReLU.
"""
import torch
import torch.nn.functional as F
from torch.nn import LeakyReLU, init
from torch.nn.modules.module import Module
from torch.nn.parameter import Parameter

import numpy as np

from . import config, ops, quant
from .molecule_gcn import RPYNQ  # noqa: F401  (same Function in both reference files)
from .pyg_lite import add_remaining_self_loops, sort_edge_index

my_ip = None
quant_constants = None        # set by init_SGRACE when the quantised path is selected
layern = 1                    # SG.py:327-359: the hardware path alternates two constant sets


def _quantised():
    return bool(config.fake_quantization) or bool(config.hardware_quantize)


def _f32_bits(x):
    return np.asarray(x, dtype=np.float32).view(np.int32).item()


def _program_quant_registers(rm, qc):
    """SG.py:334-365, :476: what FPYNQ_GAT.forward writes before AP_START."""
    rm.scale_fea = qc.scale_fea
    rm.deq_factor = _f32_bits(qc.deq_o)
    rm.quantization_scale_fea = _f32_bits(1 / qc.f_s)
    rm.quantization_scale_w = _f32_bits(1 / qc.w_s)
    rm.quantization_scale_adj = _f32_bits(1 / qc.a_s)
    rm.quantized_multiplier = qc.internal_quantization


def _torch_dtype():
    import numpy as np
    return torch.float16 if np.dtype(config.float_type) == np.dtype(np.float16) else torch.float32


def sym_norm2(edge_index, num_nodes, edge_weight=None, fill=0, dtype=None):
    """SG.py:18-51: add the missing self loops with weight `fill`, sort by (row, col),
    D^-1/2 A D^-1/2 with D = row sums."""
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=dtype, device=edge_index.device)
    edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill, num_nodes)
    edge_index, edge_weight = sort_edge_index(edge_index, edge_weight, num_nodes)
    row, col = edge_index
    deg = torch.zeros(num_nodes, dtype=edge_weight.dtype, device=edge_weight.device).index_add_(0, row, edge_weight)
    deg_inv_sqrt = deg.pow(-0.5)
    deg_inv_sqrt[deg_inv_sqrt == float('inf')] = 0
    return edge_index, deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]


def _edge_csr(adj, edge_index, norm, n, dtype):
    """The CSR the kernel reads, from the (row-sorted) COO the reference ships (SG.py:1243-1247)."""
    if isinstance(adj, ops.Csr):
        return adj.to(dtype)
    row = edge_index[0].to(torch.int32).contiguous()
    return ops.Csr.from_coo(row, edge_index[1].to(torch.int32).contiguous(), norm.to(dtype).contiguous(), n, n)


def _fq_signed(x, s, z, qbits):
    """quantization_fbits (SG.py:238-251)."""
    t = 1 / s * x + z
    if qbits == 1:
        return torch.where(t < 0, torch.full_like(t, -0.5), torch.full_like(t, 0.5))
    lim = 2 ** (qbits - 1) - 1
    return torch.clip(torch.round(t), min=-lim, max=lim) / (2 ** (qbits - 1))


def _fq_unsigned(x, s, z, qbits):
    """quantization_ufbits (SG.py:253-265)."""
    q = torch.clip(torch.round(1 / s * x + z), min=0, max=2 ** qbits - 1)
    return q / 2 if qbits == 1 else q / (2 ** (qbits - 1))


class FPYNQ_GAT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, my_ip, self, adj, nnz_adj, input, weights, attention, out_features, dropout, relu):
        ctx.nheads, ctx.alpha, ctx.relu = self.nheads, self.alpha, relu
        ctx.gat = int(config.compute_attention)
        qc = None
        if _quantised():
            if quant_constants is None:
                raise RuntimeError("config.fake_quantization is set: call init_SGRACE() with config.w_qbits in {8, 4, 2, 1}")
            qc = quant_constants
        if config.acc == 1:
            dt = _torch_dtype()
            if qc is not None:
                global layern
                if dt != torch.float32:
                    raise TypeError("the quantised layer works on float32 buffers (SG.py:1545)")
                if layern == 2:
                    qc = qc.second_layer()
                layern = 2 if layern == 1 else 1
                _program_quant_registers(my_ip.register_map, qc)
            rm = my_ip.register_map
            A = self._csr.to(dt)
            Wt = weights.detach().t().to(dt).contiguous()
            fea = input.detach()
            if int(rm.gemm_mode) == 0:
                # the CSR of a feature matrix is rebuilt only when the tensor changes (node features are fixed
                # over the epochs of a node-classification run)
                fea = ops.cached_on(input, ("fea_csr", dt), lambda: ops.Csr.from_dense(
                    fea if fea.layout == torch.strided else fea.to_dense(), dt))
            else:
                fea = fea.to(dt).contiguous()
            my_ip.alpha = self.alpha
            # config.hardware_quantize: the bitstream's own quantiser -- integer operands on the int8 matrix cores where
            # they are the faster form (dense features wider than 128 columns; sgx.h SGX_QUANT_INT8_AUTO);
            # config.fake_quantization alone: the fp32 emulation of the grid, as the reference states it
            int8 = "auto" if (qc is not None and config.hardware_quantize) else False
            if ctx.gat:
                out, E, S = my_ip.run_layer(A, fea, Wt, attention=attention.detach().to(dt).reshape(-1).contiguous(),
                                            want_edge_outputs=True, quant=qc, quant_int8=int8)
            else:
                out, E, S = my_ip.run_layer(A, fea, Wt, quant=qc, quant_int8=int8), None, None
            ctx.csr = A
            ctx.save_for_backward(input, weights, out, *([E, S] if ctx.gat else []))
            return out.float()                                            # SG.py:543 `.float()`

        # ---- no accelerator: the dense emulation of SG.py:563-681 --------------------------------
        input = input.float()
        input_q, weights_q, adj_d = input, weights, adj.to_dense()
        if qc is not None:                                                # SG.py:570-626
            input_q = _fq_unsigned(input, qc.f_s, qc.f_z, qc.w_qbits)
            weights_q = _fq_signed(weights, qc.w_s, qc.w_z, qc.w_qbits)
        Wh = torch.mm(input_q, weights_q)
        if qc is not None:
            iq = qc.internal_quantization
            Wh = Wh / (2 ** qc.scale_fea)
            Wh = torch.clip(Wh, min=-(2 ** iq - 1) / (2 ** iq), max=(2 ** iq - 1) / (2 ** iq))
            Wh = torch.round(Wh, decimals=iq - 1)
            attention = _fq_signed(attention, qc.w_s, qc.w_z, qc.w_qbits)
            adj_d = _fq_unsigned(adj_d, qc.a_s, qc.a_z, qc.w_qbits)
        Wh1 = torch.matmul(Wh, attention[:out_features, :])
        Wh2 = torch.matmul(Wh, attention[out_features:, :])
        e = self.leakyrelu(Wh1 + Wh2.T)
        attention1 = torch.where(adj_d > 0, e, -9e15 * torch.ones_like(e))
        attentions = F.softmax(attention1, dim=1)
        if ctx.gat:
            output_cpu = torch.matmul(attentions, Wh)
        else:
            output_cpu = torch.matmul(adj_d, Wh)
        if relu == 1:
            output_cpu = torch.where(output_cpu > 0, output_cpu, torch.zeros_like(output_cpu))
        if qc is not None:
            output_cpu = output_cpu * qc.deq_o                            # SG.py:666-667
        ctx.csr = None
        ctx.save_for_backward(input, weights, output_cpu, e, attentions if ctx.gat else adj_d, adj_d)
        return output_cpu

    @staticmethod
    def backward(ctx, grad_output):
        none = None
        g = grad_output.float()
        if ctx.csr is None:                                               # dense twin (SG.py:884-1126)
            input, weights, output, e, P, adj_d = ctx.saved_tensors
            Wh = input @ weights
            if ctx.gat:
                softmax_out = g @ Wh.t()
                dx = P * softmax_out
                sg = dx - P * dx.sum(dim=1, keepdim=True)
                sg = torch.where(adj_d > 0, sg, torch.zeros_like(sg))
                sg = ((e > 0) + ctx.alpha * (e <= 0)) * sg
                grad_attention = torch.cat([Wh.t() @ sg.sum(dim=1), (sg @ Wh).sum(dim=0)]).unsqueeze(1)
            else:
                grad_attention = torch.zeros((2 * weights.shape[1], 1), device=g.device)
            grad_input = P @ (g @ weights.t())
            grad_weights = input.t() @ (P @ g)
            return none, none, none, none, grad_input, grad_weights, grad_attention, none, none, none

        saved = ctx.saved_tensors
        input, weights, out = saved[0].float(), saved[1].float(), saved[2]
        if input.layout != torch.strided:
            input = input.to_dense()
        A = ctx.csr
        if ctx.gat:
            E, S = saved[3], saved[4]
            P = ops.Csr(A.rowptr, A.col, S.contiguous(), A.n_cols, A.plan if A.wants_plan else None)   # attention matrix, fp32 values; A's schedule
            Wh = ops.xw_dense(input.contiguous(), weights.t().contiguous())       # X . W, fp32 (SG.py:601); rows padded to 16 B
            sg, g1 = ops.gat_backward_edges(A, E, S, g.contiguous(), Wh, ctx.alpha)
            # column sums of sg = row sums over A^T; the transposed pattern is built once per graph
            if getattr(A, "_transpose_pattern", None) is None:
                A._transpose_pattern = ops.csr_transpose(A, return_order=True)
            AT, order = A._transpose_pattern
            ones = torch.ones((A.n_rows, 1), dtype=torch.float32, device=g.device)
            g2 = ops.spmm(ops.Csr(AT.rowptr, AT.col, sg[order].contiguous(), AT.n_cols), ones, use_plan=False)
            ga = ops.xt_g(Wh, torch.cat([g1.unsqueeze(1), g2], dim=1).contiguous())   # [F, 2] = Wh^T [g1 g2]
            grad_attention = torch.cat([ga[:, 0], ga[:, 1]]).unsqueeze(1)
        else:
            P = A.to(torch.float32)
            grad_attention = torch.zeros((2 * weights.shape[1], 1), device=g.device)
        pg = ops.spmm(P, g.contiguous())                                       # P @ g
        grad_input = ops.xw_dense(pg, weights.contiguous())                    # (P @ g) @ W^T == P @ (g @ W^T)
        if grad_input.stride(0) != grad_input.shape[1]:
            grad_input = grad_input.contiguous()
        grad_weights = ops.xt_g(input.contiguous(), pg)                        # X^T @ (P @ g)
        return none, none, none, none, grad_input, grad_weights, grad_attention, none, none, none


class Relu_SGRACE(Module):
    def __init__(self):
        super(Relu_SGRACE, self).__init__()
        self.fn = RPYNQ.apply

    def forward(self, x):
        return self.fn(x)


class GATConv_SGRACE(Module):
    """GAT / GCN layer of the SGRACE library.  `nheads` only widens W (single head, SG.py:1176-1178);
    the bias parameter exists and is never added, as in the reference."""

    def __init__(self, in_features, out_features, nheads=1, bias=True, dropout=0.2, alpha=0.2, concat=False):
        super(GATConv_SGRACE, self).__init__()
        self.in_features, self.out_features = in_features, out_features
        self.alpha, self.dropout = alpha, dropout
        self.weight = Parameter(torch.FloatTensor(in_features, out_features * nheads))
        init.xavier_uniform_(self.weight.data, gain=1.414)
        self.attention = Parameter(torch.empty(size=(2 * out_features * nheads, 1)))
        init.xavier_uniform_(self.attention.data, gain=1.414)
        self.leakyrelu = LeakyReLU(self.alpha)
        self.nheads, self.concat = nheads, concat
        self.fn = FPYNQ_GAT.apply
        self.my_ip = my_ip if config.acc == 1 else None
        self._csr = None
        if bias:
            self.bias = Parameter(torch.FloatTensor(out_features))
        else:
            self.register_parameter('bias', None)

    def run_kernel(self):
        self.my_ip.register_map.CTRL.AP_START = 1
        kernel_done = self.my_ip.register_map.CTRL.AP_DONE
        while kernel_done == 0:
            kernel_done = self.my_ip.register_map.CTRL.AP_DONE

    def forward(self, compute_attention, dense, relu, input, edge_index, norm, adj):
        nnz_adj = len(norm)
        if config.acc == 1:
            if self.my_ip is None:
                self.my_ip = my_ip
            if self.my_ip is None:
                raise RuntimeError("call init_SGRACE() before the first forward (config.acc == 1)")
            rm = self.my_ip.register_map
            rm.relu, rm.gemm_mode, rm.gat_mode = relu, dense, compute_attention
            rm.nnz_adj1 = nnz_adj
            # the CSR (with its row plan, quantised copy and dead-row check) is rebuilt only when the edge list
            # changes: the demo passes the same graph every epoch
            if isinstance(adj, ops.Csr):
                self._csr = adj.to(_torch_dtype())
            else:
                # kept on the edge list together with the `norm` it was built from (held, so that its identity
                # cannot be taken over by another tensor): rebuilt when either changes
                slot = ops.cached_on(edge_index, ("csr", input.shape[0], _torch_dtype()), dict)
                if slot.get("norm") is not norm or slot.get("norm_version") != norm._version:
                    slot.update(norm=norm, norm_version=norm._version,
                                csr=_edge_csr(None, edge_index, norm, input.shape[0], _torch_dtype()))
                self._csr = slot["csr"]
        return self.fn(self.my_ip, self, adj, nnz_adj, input, self.weight, self.attention, self.out_features,
                       self.dropout, relu)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class GAT_PYNQ(Module):
    """The two-layer model of the SGRACE demo (demo/emulation/demo_sgrace.py:271-400), same call pattern:
    sym_norm2 -> GATConv_SGRACE(compute_attention, dense=0, relu=1) -> Relu_SGRACE ->
    GATConv_SGRACE(dense=1, relu=0) -> dropout -> Linear.  The demo reads the feature / class counts from
    its global `dataset`; here they are constructor arguments."""

    def __init__(self, num_node_features, hidden_channels, head_count, num_classes):
        super(GAT_PYNQ, self).__init__()
        self.att2 = GATConv_SGRACE(num_node_features, hidden_channels, head_count, dropout=0.1, alpha=0.2, concat=False)
        self.conv22 = GATConv_SGRACE(hidden_channels * head_count, hidden_channels, 1)
        self.reluh = Relu_SGRACE()
        self.lin = torch.nn.Linear(hidden_channels, num_classes)

    def forward(self, x, edge_index):
        def normalise():                                          # once per graph, not per call
            ei, norm = sym_norm2(edge_index, x.size(0))
            adj = _edge_csr(None, ei, norm, x.size(0), _torch_dtype()) if config.acc == 1 else \
                torch.sparse_coo_tensor(ei, norm, (x.size(0), x.size(0)))
            return ei, norm, adj

        ei, norm, adj = ops.cached_on(edge_index, ("sym_norm2", x.size(0), config.acc, _torch_dtype()), normalise)
        x = self.att2(config.compute_attention, 0, 1, x, ei, norm, adj)
        x = self.reluh(x)
        x = self.conv22(config.compute_attention, 1, 0, x, ei, norm, adj)
        x = F.dropout(x.float(), p=0.5, training=self.training)
        return self.lin(x)


def init_SGRACE(device=None):
    """SG.py:1271: opens the overlay and publishes the IP handle the layers use."""
    global my_ip, quant_constants, layern
    layern = 1
    quant_constants = quant.constants(config.w_qbits) if _quantised() else None
    if config.acc == 1:
        from .pynq_shim import Overlay
        ol = Overlay("gat_all_unsigned.bit", device=device or config.device)
        my_ip = ol.mmult_top_0
        if quant_constants is not None:          # SG.py:1745-1839: f_align is the hardware's input alignment, unused here
            my_ip.register_map.beta_qu = {8: 255, 4: 15, 2: 2, 1: 1}[config.w_qbits]
            my_ip.register_map.f_align = {8: 0, 4: 4, 2: 6, 1: 7}[config.w_qbits]
    return my_ip
