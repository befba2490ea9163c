#!/usr/bin/env python3
"""Row pitch of the gathered table against the 128-byte line: the aggregation over tables of 41 / 47 / 100 fp16
columns stored with the minimal 16-byte pitch and with rows padded to whole 128-byte lines.  One JSON line per case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sgracex1_amd import graphs, ops  # noqa: E402
from tools.bench_configs import timed  # noqa: E402

dev = torch.device("cuda")


def main():
    n = 2_449_029
    A = graphs.uniform_graph(n, 123_700_000, seed=4)
    A.plan
    for width, pitches in ((47, (48, 64)), (41, (48, 64)), (100, (104, 128)), (24, (24, 32, 64)), (72, (72, 128))):
        rec = {"nodes": n, "edges": A.nnz, "width": width}
        out = torch.empty((n, width), dtype=torch.float16, device=dev)
        for pitch in pitches:
            H = torch.empty((n, pitch), dtype=torch.float16, device=dev).normal_()
            rec[f"ms_pitch_{pitch}"] = timed(lambda: ops.spmm(A, H, n_feat=width, out=out), 10)
            del H
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
