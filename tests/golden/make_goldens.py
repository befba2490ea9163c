#!/usr/bin/env python3
"""Builds the committed fixtures under tests/golden/ from the reference's DATA files.

Run in the build container only (it reads /root/reference, which does not exist on
the GPU box):   python tests/golden/make_goldens.py

What is written (data only -- inputs and the reference's own recorded outputs):

  <name>.npz                 the reference's matrices in CSR-text format
                             (gnn-rfsoc-mt-all-2022/data/matrices/*.txt, loaders
                             main_float.cpp:415-659, weights :149-200) repacked as
                             int32 / float32 arrays: adj_rowptr, adj_col, adj_val,
                             fea_rowptr, fea_col, fea_val, [fea_dense], w [M_fea,P],
                             [w2].  Values are the decimal text parsed to float32,
                             exactly what `ss >> float` yields in the testbench.
  known_answers.json         numbers the reference itself recorded for this path:
                             csim log rows 0 and 31 (citeseer, HALF build), the
                             hardware row 0 and the scipy row 0 printed in
                             jupyter/test/mmult-master.ipynb, the 4x4 hand KAT.
  mutag_raw.npz              MUTAG raw TU files (jupyter/molecule_gcn/MUTAG/raw)
                             as integer arrays, for the end-to-end molecule_gcn run.
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
MAT = os.path.join(REF, "gnn-rfsoc-mt-all-2022/data/matrices")
OUT = os.path.dirname(os.path.abspath(__file__))


def _tokens(line):
    return [t for t in re.split(r"[,\s]+", line.strip()) if t]


def read_csr_text(path):
    with open(path) as f:
        lines = [ln for ln in f.read().split("\n") if ln.strip()]
    assert len(lines) == 3, (path, len(lines))
    rowptr = np.array(_tokens(lines[0]), dtype=np.int64).astype(np.int32)
    col = np.array(_tokens(lines[1]), dtype=np.int64).astype(np.int32)
    val = np.array([float(t) for t in _tokens(lines[2])], dtype=np.float64).astype(np.float32)
    assert len(col) == len(val) == rowptr[-1], (path, len(col), len(val), rowptr[-1])
    return rowptr, col, val


def read_dense_text(path):
    rows = []
    with open(path) as f:
        for ln in f:
            if ln.strip():
                rows.append([float(t) for t in _tokens(ln)])
    return np.array(rows, dtype=np.float64).astype(np.float32)


def pack(name, adj, fea=None, w=None, w2=None, fea_dense=None):
    d = {}
    arp, aci, ava = read_csr_text(os.path.join(MAT, adj))
    d.update(adj_rowptr=arp, adj_col=aci, adj_val=ava)
    if fea:
        frp, fci, fva = read_csr_text(os.path.join(MAT, fea))
        d.update(fea_rowptr=frp, fea_col=fci, fea_val=fva)
    if fea_dense:
        d["fea_dense"] = read_dense_text(os.path.join(MAT, fea_dense))
    if w:
        d["w"] = read_dense_text(os.path.join(MAT, w))
    if w2:
        d["w2"] = read_dense_text(os.path.join(MAT, w2))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items()})


def csim_log_values():
    path = os.path.join(REF, "gnn-rfsoc-mt-all-2022/hls/gnn/solution1/gnn/solution1/csim/report/"
                             "mmult_top_csim.log")
    vals = {}
    with open(path) as f:
        for ln in f:
            m = re.match(r"out :data index= (\d+) (\d+) kernel = (\S+)", ln)
            if m:
                vals.setdefault(m.group(1), []).append(m.group(3))   # keep the printed text
    return vals


def notebook_rows():
    nb = json.load(open(os.path.join(REF, "jupyter/test/mmult-master.ipynb")))
    out = {}
    for idx, key in ((37, "hw_row0_fp16_P16"), (55, "scipy_row0_fp32_P21")):
        text = "".join(nb["cells"][idx]["outputs"][0]["text"])
        out[key] = [t for t in re.split(r"[\[\]\s]+", text) if t]
    return out


def mutag():
    raw = os.path.join(REF, "jupyter/molecule_gcn/MUTAG/raw")
    A = np.loadtxt(os.path.join(raw, "MUTAG_A.txt"), delimiter=",", dtype=np.int64).astype(np.int32)
    gi = np.loadtxt(os.path.join(raw, "MUTAG_graph_indicator.txt"), dtype=np.int64).astype(np.int32)
    gl = np.loadtxt(os.path.join(raw, "MUTAG_graph_labels.txt"), dtype=np.int64).astype(np.int32)
    nl = np.loadtxt(os.path.join(raw, "MUTAG_node_labels.txt"), dtype=np.int64).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "mutag_raw.npz"), A=A, graph_indicator=gi,
                        graph_labels=gl, node_labels=nl)
    print("mutag", A.shape, gi.shape, gl.shape, nl.shape)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures are already committed")
    pack("test", "test_adj.txt", "test_feat.txt", "test_weights.txt")
    pack("test2", "test_adj2.txt", "test_feat2.txt", "test_weights2.txt")
    pack("mol", "mol_adj.txt", "mol_feat.txt", "mol_weights.txt", fea_dense="mol_feat_dense.txt")
    pack("cora", "cora_adj.txt", "cora_feat.txt", "cora_weights.txt", "cora_weights2.txt")
    pack("citeseer", "citeseer_adj.txt", "citeseer_feat.txt", "citeseer_weights.txt",
         "citeseer_weights2.txt")
    pack("pubmed", "pubmed_adj.txt", None, "pubmed_weights.txt", "pubmed_weights2.txt")
    ka = {
        "_sources": {
            "csim_log": "gnn-rfsoc-mt-all-2022/hls/gnn/solution1/gnn/solution1/csim/report/"
                        "mmult_top_csim.log:21-62 (citeseer, gemm_mode=0, relu=0, HALF build, P_w=32 "
                        "with the 21 weight columns of citeseer_weights.txt; same numbers README.md:70-88)",
            "hw_row0_fp16_P16": "jupyter/test/mmult-master.ipynb cell 37 output (RFSoC hardware, citeseer, P_w=16)",
            "scipy_row0_fp32_P21": "jupyter/test/mmult-master.ipynb cell 55 output "
                                   "(csr(adj) @ (csr(fea) @ w), w parsed as float16)",
            "test_kat": "data/matrices/test_{adj,feat,weights}.txt, main_float.cpp:102-111: "
                        "A row0=[.5,.5,0,0], X row0=[1,2,0,0], W=e1 => D[0]=[0.5,0], other rows 0",
        },
        "csim_log": csim_log_values(),
        "test_kat": {"D": [[0.5, 0.0], [0.0, 0.0], [0.0, 0.0], [0.0, 0.0]]},
    }
    ka.update(notebook_rows())
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("known answers:", {k: (len(v) if not isinstance(v, dict) else list(v)) for k, v in ka.items()})
    mutag()


if __name__ == "__main__":
    main()
