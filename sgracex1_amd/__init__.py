"""sgracex1_amd -- MI355X-native fused GNN layer D = act(A . (X . W)) behind the interface of
hadimsnj/SGRACEx1's accelerator path (autograd Functions, modules, `relu` / `gemm_mode` flags,
a pynq-shaped register map).  Compute lives in csrc/ (hand-written HIP for gfx950, exported
through the C ABI of include/sgx.h); this package is the host side.

Importing the package does not load the HIP library; `sgracex1_amd.ops` (and everything
that runs a layer) does, and raises if it is missing.
"""
__version__ = "0.1.0"
