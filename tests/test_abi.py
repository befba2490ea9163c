"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what
include/sgx.h declares, the ctypes struct mirrors the header, and argument validation answers
with the documented status codes without touching a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sgx.h")


@pytest.fixture(scope="module")
def L():
    from sgracex1_amd import build
    build.build()
    from sgracex1_amd import _lib
    return _lib


def _header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(L):
    declared = _header_functions()
    assert declared == sorted(L.SYMBOLS)
    for name in declared:
        assert hasattr(L.lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH], text=True)
    exported = set(re.findall(r" T (sgx_[a-z0-9_]+)", out))
    assert exported == set(declared)


def test_header_compiles_as_c_and_struct_layout_matches(L, tmp_path):
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "sgx.h"\n'
        "int main(void){\n"
        ' printf("%zu\\n", sizeof(sgx_layer_desc));\n'
        + "".join(f' printf("{name} %zu\\n", offsetof(sgx_layer_desc, {name}));\n' for name, _ in L.LayerDesc._fields_)
        + ' printf("quant_struct %zu\\n", sizeof(sgx_quant));\n'
        + "".join(f' printf("q.{name} %zu\\n", offsetof(sgx_quant, {name}));\n' for name, _ in L.Quant._fields_)
        + " return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe)])
    lines = subprocess.check_output([str(exe)], text=True).split("\n")
    assert int(lines[0]) == ctypes.sizeof(L.LayerDesc)
    for ln in lines[1:]:
        if ln:
            name, off = ln.split()
            if name == "quant_struct":
                assert ctypes.sizeof(L.Quant) == int(off)
            elif name.startswith("q."):
                assert getattr(L.Quant, name[2:]).offset == int(off), name
            else:
                assert getattr(L.LayerDesc, name).offset == int(off), name


def test_version_and_status_strings(L):
    declared = int(re.search(r"#define\s+SGX_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert L.lib.sgx_version() == declared
    assert L.status_string(0) == "ok"
    for code in range(-7, 0):
        assert L.status_string(code) != "unknown status"
    assert L.status_string(-99) == "unknown status"


def test_argument_checks_need_no_gpu(L):
    d = L.LayerDesc()
    assert L.lib.sgx_layer_forward(None, None) == -1                       # SGX_ERR_NULL
    assert L.lib.sgx_layer_workspace_bytes(ctypes.byref(d)) == 0           # M_fea = 0 -> bad desc
    d.N_adj, d.M_adj, d.M_fea, d.P_w = 10, 10, 4, 8
    d.dtype = 7
    assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == -3            # SGX_ERR_UNSUPPORTED
    d.dtype = 0
    d.gemm_mode = 2                                                        # backward-offload mode: not in the public HLS
    assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == -3
    d.gemm_mode = 1
    assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == -1            # buffers missing
    # K.cpp:3876-3889: bias_count > 0 returns before computing anything
    d.bias_count = 3
    assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == 0
    d.bias_count = 0
    need = L.lib.sgx_layer_workspace_bytes(ctypes.byref(d))
    assert need >= 10 * 8 * 2 and need % 256 == 0
    assert L.lib.sgx_spmm_csr(0, 0, 1, 0, 4, 4, 0, None, None, None, None, 8, None, 8, None, None, 0, None) == -2
    assert L.lib.sgx_spmm_csr(0, 0, 1, 0, 4, 4, 8, None, None, None, None, 8, None, 8, None, None, 0, None) == -1
    assert L.lib.sgx_xw_dense(0, 0, 1, 4, 0, 8, None, 8, None, 8, None, 8, None) == -2
    assert L.lib.sgx_transpose(0, 4, 4, None, 4, None, 4, None) == -1
    # aggregate-first order: dense X, GCN aggregate, default arithmetic only; its workspace holds Z = A.X [N_adj][M_fea]
    d.order = 1
    swapped = L.lib.sgx_layer_workspace_bytes(ctypes.byref(d))
    assert 10 * 8 * 2 <= swapped <= need and swapped % 256 == 0
    for field, bad in (("gemm_mode", 0), ("gat_mode", 1), ("acc_mode", 1)):
        keep = getattr(d, field)
        setattr(d, field, bad)
        assert L.lib.sgx_layer_workspace_bytes(ctypes.byref(d)) == 0, field
        assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == -3, field
        setattr(d, field, keep)
    d.order = 2
    assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == -3
    d.order = 0
    # quantised layer: fp32 only, zero points of the CSR operands must be 0, bit widths bounded
    q = L.Quant()
    q.qbits, q.scale_fea, q.internal_bits = 8, 4, 16
    d.quant = ctypes.pointer(q)
    assert L.lib.sgx_layer_workspace_bytes(ctypes.byref(d)) == 0           # dtype F16 with quant
    d.dtype = 1
    q.nnz_adj = 100
    assert L.lib.sgx_layer_workspace_bytes(ctypes.byref(d)) >= need + 8 * 4 * 4 + 10 * 4 * 4 + 400
    q.zero_adj = 1.0
    assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == -3
    q.zero_adj, q.qbits = 0.0, 40
    assert L.lib.sgx_layer_forward(ctypes.byref(d), None) == -3
    assert L.lib.sgx_fake_quantize(1, 8, 1.0, 0.0, -1, None, None, None) == -2
    assert L.lib.sgx_fake_quantize(1, 0, 1.0, 0.0, 4, None, None, None) == -3
    assert L.lib.sgx_fake_quantize(1, 8, 1.0, 0.0, 4, None, None, None) == -1
    assert L.lib.sgx_requantize(4, 8, 4, None, 4, 16, None) == -2


def test_product_path_has_no_oracle_or_cpu_fallback():
    """Nothing under sgracex1_amd/ may import or link the oracle."""
    pkg = os.path.join(ROOT, "sgracex1_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower(), (fn, "mentions the oracle")
                assert "liborc" not in text
    # examples/ and tools/ neither: the only importers outside tests/ are bench.py's cpu_baseline leg and
    # __graft_entry__ (build of the checker, smoke's check)
    for sub in ("examples", "tools"):
        for fn in os.listdir(os.path.join(ROOT, sub)):
            if fn.endswith((".py", ".cpp", ".sh")):
                text = open(os.path.join(ROOT, sub, fn)).read()
                assert "from oracle" not in text and "import oracle" not in text and "oracle/" not in text, fn


def test_graft_entry_build_check_passes():
    """`__graft_entry__.build()` is the driver's does-it-build step: run it (the objects are already
    compiled, so this relinks and re-checks) -- a stale version assert there once went unnoticed."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    entry = importlib.import_module("__graft_entry__")
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "def build()" in src and "def smoke()" in src
    # everything build() asserts after compiling, without forcing a full recompile here
    from sgracex1_amd import _lib
    declared = int(re.search(r"#define\s+SGX_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert _lib.lib.sgx_version() == declared
    assert callable(entry.build) and callable(entry.smoke)


def test_package_fails_loudly_without_the_library(tmp_path):
    """No HIP library -> importing the accelerated path raises; there is no CPU or torch fallback."""
    import sys
    env = dict(os.environ, SGX_LIB_PATH=str(tmp_path / "no_such_libsgx.so"))
    out = subprocess.run([sys.executable, "-c", "import sgracex1_amd.ops"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert out.returncode != 0
    assert "ImportError" in out.stderr and "no other backend" in out.stderr
    # a file that is not a loadable library fails too (OSError from the loader), it is not skipped
    bogus = tmp_path / "libsgx.so"
    bogus.write_bytes(b"not an ELF file")
    out = subprocess.run([sys.executable, "-c", "import sgracex1_amd.ops"], cwd=ROOT, env=dict(env, SGX_LIB_PATH=str(bogus)),
                         capture_output=True, text=True)
    assert out.returncode != 0 and "Error" in out.stderr


def test_integration_md_stub_matches_the_header(L):
    """The ctypes stub printed in INTEGRATION.md is field for field the struct of include/sgx.h."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("class sgx_layer_desc(ctypes.Structure)"):]
    block = block[:block.index("\n\nsgx.sgx_layer_workspace_bytes")]
    scope = {"ctypes": ctypes}
    exec(block, scope)                                                 # defines sgx_layer_desc from the document
    doc = scope["sgx_layer_desc"]
    assert ctypes.sizeof(doc) == ctypes.sizeof(L.LayerDesc)
    assert [n for n, _ in doc._fields_] == [n for n, _ in L.LayerDesc._fields_]
    for name, _ in doc._fields_:
        assert getattr(doc, name).offset == getattr(L.LayerDesc, name).offset, name
