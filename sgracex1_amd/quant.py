"""Quantisation constants of the SGRACE bitstream (SG.py:95-174, :1298-1537, :1645-1848).

`init_SGRACE` of the reference derives, from `config.w_qbits`, the value ranges of weights (signed),
adjacency and features (unsigned), their scales and zero points, the shift applied to H = X.W
(`scale_fea`), the width of the internal pipeline (`internal_quantization`) and the factor that
brings the aggregate back to real units (`deq_o`).  `constants(w_qbits)` returns the same numbers;
`QuantConstants.as_struct()` is the `sgx_quant` the C ABI takes.

The ranges are the ones live (not commented out) in the reference for each bit width; they are
data the reference tuned per dataset, so every field can be overridden.
"""
from dataclasses import dataclass, replace

from . import _lib

# w_qbits -> (w_min, w_max, a_min, a_max, f_min, f_max, w_min2, w_max2, f_min2, f_max2, go_min, go_max)
#            SG.py:1327-1345 (8), :1451-1466 (4), :1509-1520 (2), :1527-1537 (1)
_RANGES = {
    8: (-1.0, 1.0, 0.0, 1.0, 0.0, 1.0, -1.0, 1.0, 0.0, 1.0, -0.10, 0.10),
    4: (-1.0, 1.0, 0.0, 1.0, 0.0, 1.0, -1.0, 1.0, 0.0, 1.0, -0.10, 0.10),
    2: (-0.1, 0.1, 0.0, 0.1, 0.0, 1.0, -0.1, 0.1, 0.0, 1.0, -0.10, 0.10),
    1: (-0.1, 0.1, 0.0, 0.1, 0.0, 1.0, -0.1, 0.1, 0.0, 1.0, -0.10, 0.10),
}
# w_qbits -> (scale_fea, scale_fea2, deq_o multiplier, internal_quantization)   SG.py:1698-1840
_PIPELINE = {8: (4, 4, 2.0, 16), 4: (3, 3, 2.0, 8), 2: (3, 3, 2.0, 4), 1: (2, 2, 2.0, 4)}


def _affine(alpha, beta, alpha_q, beta_q, w_qbits):
    """generate_quantization_constants (SG.py:95-132): (s_o, s, z)."""
    shift = 2 ** 2 if w_qbits == 1 else 2 ** w_qbits
    beta_o, alpha_o = beta_q / shift, alpha_q / shift
    s_o = (beta - alpha) / (beta_o - alpha_o)
    s = (beta - alpha) / (beta_q - alpha_q)
    z = int((beta * alpha_q - alpha * beta_q) / (beta - alpha))
    return s_o, s, z


def signed_constants(alpha, beta, qbits, w_qbits=None):
    """generate_quantization_qbits_constants (SG.py:152-174)."""
    if qbits == 1:
        alpha_q, beta_q = -1, 1
    else:
        alpha_q, beta_q = -2 ** (qbits - 1) + 1, 2 ** (qbits - 1) - 1
    return _affine(alpha, beta, alpha_q, beta_q, qbits if w_qbits is None else w_qbits)


def unsigned_constants(alpha, beta, qbits, w_qbits=None):
    """generate_quantization_uqbits_constants (SG.py:135-149)."""
    return _affine(alpha, beta, 0, 2 ** qbits - 1, qbits if w_qbits is None else w_qbits)


@dataclass(frozen=True)
class QuantConstants:
    w_qbits: int
    w_s: float
    w_z: int
    a_s: float
    a_z: int
    f_s: float
    f_z: int
    scale_fea: int
    internal_quantization: int
    deq_o: float
    # second layer of the hardware path (SG.py:349-358)
    w_s2: float = 0.0
    w_z2: int = 0
    f_s2: float = 0.0
    f_z2: int = 0
    scale_fea2: int = 0
    deq_o2: float = 0.0
    # gradient scales the reference computes and never applies in the forward path (SG.py:1690-1691)
    deq_gw: float = 0.0
    deq_gi: float = 0.0

    def second_layer(self):
        """The constants the hardware path switches to for the second layer (`layern == 2`)."""
        return replace(self, w_s=self.w_s2, w_z=self.w_z2, f_s=self.f_s2, f_z=self.f_z2, scale_fea=self.scale_fea2,
                       deq_o=self.deq_o2)

    def as_struct(self, nnz_adj, nnz_fea=0, adj_done=False):
        q = _lib.Quant()
        q.qbits, q.scale_fea, q.internal_bits = self.w_qbits, self.scale_fea, self.internal_quantization
        q.flags = _lib.SGX_QUANT_ADJ_DONE if adj_done else 0
        q.inv_scale_fea, q.zero_fea = 1 / self.f_s, self.f_z
        q.inv_scale_w, q.zero_w = 1 / self.w_s, self.w_z
        q.inv_scale_adj, q.zero_adj = 1 / self.a_s, self.a_z
        q.deq_factor = self.deq_o
        q.nnz_adj, q.nnz_fea = int(nnz_adj), int(nnz_fea)
        return q


def constants(w_qbits, **override):
    """The constants `init_SGRACE` publishes for config.w_qbits (SG.py:1645-1848)."""
    if w_qbits not in _RANGES:
        raise ValueError("w_qbits must be 8, 4, 2 or 1 (SG.py:1298, :1437, :1503, :1524)")
    w_min, w_max, a_min, a_max, f_min, f_max, w_min2, w_max2, f_min2, f_max2, go_min, go_max = _RANGES[w_qbits]
    scale_fea, scale_fea2, boost, internal = _PIPELINE[w_qbits]
    w_s_o, w_s, w_z = signed_constants(w_min, w_max, w_qbits)
    w_s_o2, w_s2, w_z2 = signed_constants(w_min2, w_max2, w_qbits)
    a_s_o, a_s, a_z = unsigned_constants(a_min, a_max, w_qbits)
    f_s_o, f_s, f_z = unsigned_constants(f_min, f_max, w_qbits)
    f_s_o2, f_s2, f_z2 = unsigned_constants(f_min2, f_max2, w_qbits)
    go_s_o, _go_s, _go_z = unsigned_constants(go_min, go_max, 8, w_qbits)      # go_qbits = 8 (SG.py:1647)
    c = QuantConstants(w_qbits=w_qbits, w_s=w_s, w_z=w_z, a_s=a_s, a_z=a_z, f_s=f_s, f_z=f_z, scale_fea=scale_fea,
                       internal_quantization=internal, deq_o=w_s_o * f_s_o * a_s_o * boost,
                       w_s2=w_s2, w_z2=w_z2, f_s2=f_s2, f_z2=f_z2, scale_fea2=scale_fea2,
                       deq_o2=w_s_o2 * f_s_o2 * a_s_o * boost,
                       deq_gw=f_s_o * a_s_o * go_s_o, deq_gi=a_s_o * go_s_o * w_s_o)
    return replace(c, **override) if override else c
