#!/usr/bin/env python3
"""PCIe-inclusive rate of the pynq-shaped compatibility path (host buffers -> HBM -> kernel -> host)
next to the device-resident rate of the same layer: what the register-map flow costs when the
CSR arrays live in host memory, as they do under PYNQ.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sgracex1_amd import graphs, ops, pynq_shim  # noqa: E402

n, edges, f_in, P = 1 << 20, 16_000_000, 64, 64
A = graphs.uniform_graph(n, edges, seed=1)
X = torch.rand((n, f_in), device="cuda").half()
Wt = ((torch.rand((P, f_in), device="cuda") - 0.5) / 4).half()
A.plan
out = torch.empty((n, P), device="cuda", dtype=torch.float16)
for _ in range(3):
    ops.layer_forward(A, X, Wt, relu=True, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ops.layer_forward(A, X, Wt, relu=True, out=out)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 10

ip = pynq_shim.Overlay("gnn_all.bit").mmult_top_0
alloc = pynq_shim.allocate
bufs = dict(rp=alloc(n + 1, np.int32), ci=alloc(A.nnz, np.int32), va=alloc(A.nnz, np.float16),
            x=alloc(n * f_in, np.float16), B=alloc(P * f_in, np.float16), D=alloc(n * P, np.float16))
bufs["rp"][:] = A.rowptr.cpu().numpy()
bufs["ci"][:] = A.col.cpu().numpy()
bufs["va"][:] = A.val.cpu().numpy()
bufs["x"][:] = X.cpu().numpy().reshape(-1)
bufs["B"][:] = Wt.cpu().numpy().reshape(-1)
rm = ip.register_map
rm.N_adj = rm.M_adj = n
rm.M_fea, rm.P_w, rm.relu, rm.gemm_mode = f_in, P, 1, 1
rm.rowPtr_adj1_offset_1 = bufs["rp"].physical_address
rm.columnIndex_adj1_offset_1 = bufs["ci"].physical_address
rm.values_adj1_offset_1 = bufs["va"].physical_address
rm.values_fea1_offset_1 = bufs["x"].physical_address
rm.B_offset_1 = bufs["B"].physical_address
rm.D1_offset_1 = bufs["D"].physical_address
def start():
    rm.CTRL.AP_START = 1
    while rm.CTRL.AP_DONE == 0:
        pass


def timed_start():
    before = dict(ip.transfer_stats)
    t0 = time.perf_counter()
    start()
    dt = time.perf_counter() - t0
    return dt, {k: ip.transfer_stats[k] - before[k] for k in before}


# first start: every buffer crosses PCIe (pinned memory, asynchronous copies) and the adjacency gets its row plan;
# second start with nothing changed: the mirrors are current, only D comes back;
# a new layer on the same graph (X and W rewritten, as between the two layers of GCN_PYNQ): the adjacency stays in HBM
t_first, moved_first = timed_start()
t_same, moved_same = timed_start()
bufs["x"][:] = bufs["x"][::-1]
bufs["B"][:] = -np.asarray(bufs["B"])
t_newx, moved_newx = timed_start()
bufs["x"][:] = X.cpu().numpy().reshape(-1)
bufs["B"][:] = Wt.cpu().numpy().reshape(-1)
start()
assert np.array_equal(np.asarray(bufs["D"]).reshape(n, P), out.cpu().numpy())
adj_bytes = (n + 1) * 4 + A.nnz * 6
host_bytes = adj_bytes + n * f_in * 2 + P * f_in * 2 + n * P * 2
print(json.dumps({"nodes": n, "edges": A.nnz, "f_in": f_in, "P": P,
                  "device_resident_ms": t_dev * 1e3, "device_resident_edges_per_s": A.nnz / t_dev,
                  "host_buffers_first_start_ms": t_first * 1e3, "first_start_bytes": moved_first,
                  "first_start_GBps": (moved_first["uploaded_bytes"] + moved_first["downloaded_bytes"]) / t_first / 1e9,
                  "host_buffers_unchanged_ms": t_same * 1e3, "unchanged_bytes": moved_same,
                  "host_buffers_new_features_same_graph_ms": t_newx * 1e3, "new_features_bytes": moved_newx,
                  "adjacency_bytes": adj_bytes, "all_buffers_bytes": host_bytes,
                  "round2_same_layer_ms": 39.0,
                  "note": "round 2 copied every buffer through pageable memory on every AP_START: 39 ms for this layer"}))
