"""Loaders for the committed fixtures under tests/golden/ (see make_goldens.py)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (N, M_fea, P) exactly as the reference's testbench / notebooks configure them
# (main_float.cpp:40-111, mmult-master.ipynb cells 4-6)
SHAPES = {
    "test": (4, 4, 2),
    "test2": (4, 4, 2),
    "mol": (2273, 7, 64),
    "cora": (2708, 1433, 64),
    "citeseer": (3327, 3703, 21),
}


def load(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    d = {k: g[k] for k in g.files}
    if name in SHAPES:
        N, M, P = SHAPES[name]
        d["N"], d["M_fea"], d["P"] = N, M, P
        d["w"] = np.ascontiguousarray(d["w"][:M, :P])          # loaders read M_fea lines only
        d["Wt"] = np.ascontiguousarray(d["w"].T)              # [P][M_fea], what goes into B
        d["adj"] = (d["adj_rowptr"], d["adj_col"], d["adj_val"])
        d["fea"] = (d["fea_rowptr"], d["fea_col"], d["fea_val"])
    return d


def known_answers():
    with open(os.path.join(GOLD, "known_answers.json")) as f:
        return json.load(f)


# The csim log's two entries that the checked-in kernel source does not print (tests/csim_residual.py,
# profiles/r02_csim_residual.txt): one binary16 ulp each, the model's magnitude above the log's.  (row, col) -> what the
# reference-half arithmetic prints there.  Every other of the 42 logged values is reproduced to the printed digit.
CSIM_RESIDUAL = {(0, 10): "0.0995483", (31, 18): "-0.0348816"}


def assert_prints_csim_log(printed):
    """printed: {(row, col): text} for rows 0 and 31 of the citeseer layer in reference arithmetic (SPMM_BLOCK 4).
    Exactly the known two entries differ from the log, by one ulp, with exactly the known text."""
    ka = known_answers()["csim_log"]
    diff = {(int(r), j): printed[(int(r), j)] for r in ("0", "31") for j, t in enumerate(ka[r]) if printed[(int(r), j)] != t}
    assert diff == CSIM_RESIDUAL, diff
    for (r, j), text in diff.items():
        assert half_ulp_distance(np.float16(float(text)), np.float16(float(ka[str(r)][j]))) == 1
        assert abs(float(text)) > abs(float(ka[str(r)][j]))


def csr_to_dense(csr, shape):
    rp, ci, va = csr
    out = np.zeros(shape, dtype=np.float32)
    for r in range(shape[0]):
        for e in range(rp[r], rp[r + 1]):
            out[r, ci[e]] += va[e]
    return out


def half_ulp_distance(a, b):
    """|a-b| in units of binary16 ulps (monotone integer mapping of the bit patterns)."""
    def key(x):
        u = np.asarray(x, dtype=np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u)
    return np.abs(key(a) - key(b))


def sample_rows(A, rows):
    """(rowptr, compact col, val, unique table rows) of the given rows of A, on the host."""
    import torch
    rp = A.rowptr
    starts, ends = rp[rows].long(), rp[rows + 1].long()
    deg = ends - starts
    off = torch.repeat_interleave(starts - torch.cumsum(deg, 0) + deg, deg)
    idx = off + torch.arange(int(deg.sum()), device=rp.device)
    col, val = A.col[idx].long(), A.val[idx]
    uniq, inv = torch.unique(col, return_inverse=True)
    srp = torch.zeros(rows.numel() + 1, dtype=torch.int32)
    srp[1:] = torch.cumsum(deg, 0).cpu().to(torch.int32)
    return srp.numpy(), inv.cpu().numpy().astype(np.int32), val.float().cpu().numpy(), uniq
