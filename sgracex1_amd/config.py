"""Run-time flag module, same names as the reference's demo/*/config.py (emulation/config.py:1-31).

Only the flags that reach the hot path are live here; the quantisation flags of the SGRACE
bitstream (fake_quantization, hardware_quantize, w_qbits) are accepted for source compatibility
but must stay at their "off" values -- the quantised kernel is outside the fp16/fp32 path this
package implements.
"""
import numpy as np

device = "cuda"              # reference default "cpu"; the accelerator here is the GPU itself
hidden_channels = 16
layer_count = 1
load_weights = 1

accb = 0                     # backward on the accelerator (the reference's gemm_mode=2 bitstream path): not offered
acc = 1                      # 1: forward through the HIP kernels; 0: plain torch `adj @ x @ W` (the parity twin)
show_max_min = 0
min_output = 1
profiling = 0
fake_quantization = 0        # reference default 1 (emulates the int8 bitstream); unsupported here
hardware_quantize = 0
compute_attention = 0        # 0: GCN aggregate, 1: GAT edge softmax (register gat_mode)
stream_mode = 0
head_count = 1               # "not in use" in the reference too (emulation/config.py:18)

N_adj = 20480
M_adj = 20480
M_fea = 2048
P_w = hidden_channels
NNZ_adj = 1000000
NNZ_fea = 4000000
w_qbits = 32

float_type = np.float32      # element type of the accelerator buffers (SG.py:1545); np.float16 selects the HALF build's type
