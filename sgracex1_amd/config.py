"""Run-time flags of the SGRACE host library, exposed under the names the reference's per-board
`config.py` modules use (demo/emulation/config.py, demo/{rfsoc42,ultra96,zcu104}/config.py), so
that code written against `import config; config.acc = 1` keeps working:

    import sgracex1_amd.config as config

Each flag is declared once in `_FLAGS` below with what it means HERE; the module then publishes
them as plain module attributes.  Only the flags that reach the GPU hot path are live.

Defaults differ from the reference's shipped board files in one respect: those run the quantised
bitstream (`fake_quantization = 1`, `hardware_quantize = 1`, `w_qbits = 8` on the boards, `1` in
demo/emulation/config.py), here the layers are unquantised until these flags are set -- the fp16 /
fp32 layer is the path this package is about.  `reference_board_defaults()` applies the boards' values.
"""
import numpy as _np

_FLAGS = {
    # -- where and how a layer runs ------------------------------------------------------------
    "device": ("cuda", "torch device of the tensors; the accelerator is the GPU itself"),
    "acc": (1, "forward path: 1 = HIP kernels, 0 = the dense torch formulation (parity twin)"),
    "accb": (0, "backward offload register path of the GAT bitstream (gemm_mode 2): not offered; "
                "backward always runs on the device kernels"),
    "compute_attention": (0, "0 = GCN aggregate A.H, 1 = single-head GAT edge softmax (register gat_mode)"),
    "float_type": (_np.float32, "element type of the layer buffers; np.float16 selects the HALF build's type"),
    "layer_order": ("reference", "'reference': every layer forms X.W first, as the FPGA dataflow does; 'auto': a layer with "
                                 "dense features narrower than its output aggregates first, act((A.X).W) -- same sums, "
                                 "other association (sgx_layer_desc.order); no flag of this name in the reference"),
    "hidden_channels": (16, "default hidden width of the demo models"),
    "head_count": (1, "accepted, unused (the reference marks it 'not in use' as well)"),
    # -- accepted, no effect on this path --------------------------------------------------------
    "layer_count": (1, "layers per hardware call on the FPGA"),
    "load_weights": (1, "FPGA weight preload switch"),
    "stream_mode": (0, "FPGA streaming I/O switch"),
    "profiling": (0, "print per-layer host timings"),
    "show_max_min": (0, "print value ranges"),
    "min_output": (1, "quiet mode"),
    # -- quantised bitstream (SG.py:570-667, :1645-1848; sgracex1_amd/quant.py) ------------------------
    "fake_quantization": (0, "1 = layers use the quantised arithmetic of the SGRACE bitstream (fp32 tensors, "
                             "w_qbits in {8, 4, 2, 1}); read by init_SGRACE and FPYNQ_GAT"),
    "hardware_quantize": (0, "the bitstream's own quantiser: the arithmetic of fake_quantization with X and W as integer codes "
                             "on the int8 matrix cores where that is the faster form (dense features wider than 128 "
                             "columns; exact int32 sums instead of fp32 ones)"),
    "w_qbits": (32, "operand bits of the quantised layer; 32 = not quantised"),
    # -- buffer capacities of the PYNQ allocation step (informational here) ----------------------
    "N_adj": (20480, "max nodes"), "M_adj": (20480, "max nodes"), "M_fea": (2048, "max input features"),
    "NNZ_adj": (1_000_000, "max adjacency non-zeros"), "NNZ_fea": (4_000_000, "max feature non-zeros"),
}

globals().update({name: default for name, (default, _doc) in _FLAGS.items()})
P_w = hidden_channels  # noqa: F821  (published by the line above)


def reference_board_defaults(w_qbits_value=8):
    """The values of demo/{rfsoc42,ultra96,zcu104}/config.py: accelerator on, quantised arithmetic, float32 buffers."""
    g = globals()
    g.update(acc=1, accb=0, fake_quantization=1, hardware_quantize=1, w_qbits=w_qbits_value, float_type=_np.float32)


def snapshot():
    """every flag's current value (for code that changes flags for a while: `saved = config.snapshot()` ... `config.restore(saved)`)"""
    return {name: globals()[name] for name in _FLAGS}


def restore(saved):
    globals().update(saved)


def describe():
    """name -> (current value, meaning), for notebooks that want to print the configuration."""
    return {name: (globals()[name], doc) for name, (_d, doc) in _FLAGS.items()}
