"""Replaying a fixed sequence of layer launches from a hipGraph.

The molecule and Cora sized layers run in a few microseconds each, so a forward pass is bound by
launch overhead (MUTAG batch 0.031 ms eager, 0.024 ms replayed).  The C ABI never
synchronises, allocates or frees, so a sequence of calls on one stream can be captured once and
replayed; inputs are refreshed by copying into the tensors the capture saw.

    run = Graphed(lambda: model_forward(x_static))      # warm-up run + capture
    x_static.copy_(new_x); out = run()                  # replay; `out` is the captured output tensor

What has to happen BEFORE the capture is done by the warm-up call inside `Graphed`: row plans
(`Csr.plan`, one device->host copy each), the dead-row check of a GAT adjacency, workspace growth.
"""
import torch


class Graphed:
    def __init__(self, fn, warmup=2):
        self._fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # plans, workspaces and caches are created here
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
