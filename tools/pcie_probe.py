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
rm.CTRL.AP_START = 1
t0 = time.perf_counter()
for _ in range(3):
    rm.CTRL.AP_START = 1
    while rm.CTRL.AP_DONE == 0:
        pass
t_host = (time.perf_counter() - t0) / 3
assert np.array_equal(np.asarray(bufs["D"]).reshape(n, P), out.cpu().numpy())
host_bytes = (n + 1) * 4 + A.nnz * 6 + n * f_in * 2 + P * f_in * 2 + n * P * 2
print(json.dumps({"nodes": n, "edges": A.nnz, "f_in": f_in, "P": P,
                  "device_resident_ms": t_dev * 1e3, "device_resident_edges_per_s": A.nnz / t_dev,
                  "host_buffers_ms": t_host * 1e3, "host_buffers_edges_per_s": A.nnz / t_host,
                  "host_bytes_moved": host_bytes, "host_path_GBps": host_bytes / t_host / 1e9}))
