#!/usr/bin/env python3
"""Where the host time of one molecule_gcn training step goes (188-graph MUTAG batch: every kernel is a few
microseconds, the step is host-bound): cProfile over 200 steps, the top functions by own time, and the step's wall time.

    python3 tools/mutag_step_profile.py [--steps 200] [--no-profile]
"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()
    from _fixtures import GOLD
    from sgracex1_amd import molecule_gcn as MG, pyg_lite as G, pynq_shim
    dev = torch.device("cuda")
    raw = np.load(os.path.join(GOLD, "mutag_raw.npz"))
    b = G.collate(G.load_tu_raw(raw["A"], raw["graph_indicator"], raw["graph_labels"], raw["node_labels"])).to(dev)
    model = MG.GCN_PYNQ(64, 7, 2, pynq_shim.Overlay("gnn_all.bit").mmult_top_0).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    crit = torch.nn.CrossEntropyLoss()

    def train_step():
        opt.zero_grad()
        crit(model(1, b.x, b.edge_index, b.batch), b.y).backward()
        opt.step()

    for _ in range(20):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train_step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    # forward only, backward only (host time, synchronised at the end of each batch of steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        with torch.no_grad():
            model(1, b.x, b.edge_index, b.batch)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / args.steps * 1e3
    print(json.dumps({"ms_train_step_wall": round(wall, 4), "ms_forward_no_grad_wall": round(fwd, 4), "steps": args.steps}), flush=True)
    if args.no_profile:
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        train_step()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue())


if __name__ == "__main__":
    main()
