#!/usr/bin/env python3
"""Node classification with the SGRACE library's layers on the GPU kernels -- the call pattern of
the reference's demo (demo/emulation/demo_sgrace.py: init_SGRACE, GAT_PYNQ, train / test), on a
synthetic planted-partition graph because the demo's datasets (Planetoid Cora, Amazon Photo) are
downloaded by torch_geometric and are not available offline.

    python examples/sgrace_node_classification.py [--attention] [--qbits 8] [--epochs 60] [--acc 0]

--attention  GAT edge softmax instead of the GCN aggregate (config.compute_attention)
--qbits B    run the layers with the quantised arithmetic of the SGRACE bitstream (config.fake_quantization and
             config.hardware_quantize, as the reference's board configs set them: integer operands on the int8 matrix
             cores for the dense-feature layer when it is wider than 128 columns -- --hidden 256 makes layer 2 such a layer)
--emulate    with --qbits: config.fake_quantization only (the fp32 emulation of the grid everywhere)
--acc 0      the reference's dense torch emulation instead of the kernels (small graphs only)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def planted_partition(n, classes, f_in, p_in, p_out, seed, device):
    """Labels, an undirected edge list denser inside classes, and non-negative features in [0, 1]
    (the range the quantiser tables assume) that carry a weak class signal."""
    g = torch.Generator().manual_seed(seed)
    y = torch.randint(0, classes, (n,), generator=g)
    same = y[:, None] == y[None, :]
    prob = torch.where(same, torch.tensor(p_in), torch.tensor(p_out))
    upper = torch.triu(torch.rand((n, n), generator=g) < prob, diagonal=1)
    src, dst = upper.nonzero(as_tuple=True)
    edge_index = torch.stack([torch.cat([src, dst]), torch.cat([dst, src])])
    proto = (torch.rand((classes, f_in), generator=g) < 0.15).float()
    x = ((torch.rand((n, f_in), generator=g) < 0.04).float() + proto[y] * (torch.rand((n, f_in), generator=g) < 0.25)).clamp(0, 1)
    return x.to(device), edge_index.to(device), y.to(device)


def run(attention=False, qbits=32, epochs=60, acc=1, n=3000, hidden=16, seed=1, verbose=True, emulate=False):
    from sgracex1_amd import config, sgrace
    config.acc = acc
    config.compute_attention = int(attention)
    config.fake_quantization = int(qbits != 32)
    config.hardware_quantize = int(qbits != 32 and not emulate)
    config.w_qbits = qbits
    config.float_type = np.float32
    device = torch.device("cuda" if acc == 1 else "cpu")
    config.device = str(device)
    sgrace.init_SGRACE()
    torch.manual_seed(seed)
    x, edge_index, y = planted_partition(n, 5, 200, 0.02, 0.002, seed, device)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).to(device)
    train, test = perm[: n // 5], perm[n // 5:]
    model = sgrace.GAT_PYNQ(x.shape[1], hidden, 1, 5).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    crit = torch.nn.CrossEntropyLoss()
    t0 = time.time()
    for epoch in range(epochs):
        model.train()
        opt.zero_grad()
        loss = crit(model(x, edge_index)[train], y[train])
        loss.backward()
        opt.step()
        if verbose and (epoch + 1) % 20 == 0:
            print(f"epoch {epoch + 1:3d}  loss {float(loss):.4f}", flush=True)
    if device.type == "cuda":
        torch.cuda.synchronize()
    elapsed = time.time() - t0
    model.eval()
    with torch.no_grad():
        pred = model(x, edge_index).argmax(1)
    result = {"train_acc": float((pred[train] == y[train]).float().mean()),
              "test_acc": float((pred[test] == y[test]).float().mean()),
              "edges": int(edge_index.shape[1]), "ms_per_epoch": 1000 * elapsed / epochs}
    if verbose:
        print(result)
    return result, model, (x, edge_index, y)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--attention", action="store_true")
    ap.add_argument("--qbits", type=int, default=32, choices=[32, 8, 4, 2, 1])
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--acc", type=int, default=1, choices=[0, 1])
    ap.add_argument("--hidden", type=int, default=16)
    ap.add_argument("--nodes", type=int, default=3000)
    ap.add_argument("--emulate", action="store_true")
    a = ap.parse_args()
    run(a.attention, a.qbits, a.epochs, a.acc, n=a.nodes, hidden=a.hidden, emulate=a.emulate)
