#!/usr/bin/env python3
"""A 47-column fp16 table with unpadded rows (94 bytes, odd halves): gathered one element per lane by the library
as it stands against ops.spmm's copy into padded rows + 16-byte gathers.  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sgracex1_amd import graphs, ops  # noqa: E402
from sgracex1_amd._lib import check, lib  # noqa: E402
from tools.bench_configs import timed  # noqa: E402


def main():
    n, P = 2_449_029, 47
    A = graphs.uniform_graph(n, 123_700_000, seed=4)
    plan = A.plan
    H = torch.empty((n, P), dtype=torch.float16, device="cuda").normal_()
    out = torch.empty((n, P), dtype=torch.float16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def raw():
        check(lib.sgx_spmm_csr(0, 0, 1, 0, n, n, P, A.rowptr.data_ptr(), A.col.data_ptr(), A.val.data_ptr(), H.data_ptr(), P,
                               out.data_ptr(), P, plan.handle, None, 0, stream), "sgx_spmm_csr")

    rec = {"nodes": n, "edges": A.nnz, "width": P}
    rec["ms_one_element_per_lane"] = timed(raw, 5)
    ref = out.clone()
    rec["ms_ops_spmm_copy_then_vector_gathers"] = timed(lambda: ops.spmm(A, H, out=out), 5)
    rec["same_bits"] = bool(torch.equal(ref, out))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
