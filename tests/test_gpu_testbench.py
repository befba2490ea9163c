"""The C++ host (examples/sgx_testbench.cpp, the role of the reference's main_float.cpp) over the
C ABI: citeseer matrices written in the reference's text format, the kernel in the reference's
half arithmetic with SPMM_BLOCK = 4, and the printed `out :data index= i j kernel = v` lines laid
beside the lines the reference's own C simulation logged (mmult_top_csim.log:21-62)."""
import os
import re
import subprocess

import numpy as np
import pytest

from _fixtures import assert_prints_csim_log, known_answers, load

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_csr(path, csr):
    rp, ci, va = csr
    with open(path, "w") as f:
        f.write(", ".join(str(int(v)) for v in rp) + "\n")
        f.write(", ".join(str(int(v)) for v in ci) + "\n")
        f.write(", ".join("%.9g" % v for v in va) + "\n")          # 9 digits round-trip a float32


def _write_rows(path, mat):
    with open(path, "w") as f:
        for row in mat:
            f.write(", ".join("%.9g" % v for v in row) + "\n")


@pytest.fixture(scope="module")
def bench_exe():
    from sgracex1_amd import build
    return build.build_testbench()


def _run(exe, *args):
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    vals = {}
    for m in re.finditer(r"out :data index= (\d+) (\d+) kernel = (\S+)", out.stdout):
        vals[(int(m.group(1)), int(m.group(2)))] = m.group(3)
    return vals, out.stdout


def test_cpp_host_prints_the_csim_log(bench_exe, tmp_path):
    d = load("citeseer")
    _write_csr(tmp_path / "adj.txt", d["adj"])
    _write_csr(tmp_path / "fea.txt", d["fea"])
    _write_rows(tmp_path / "w.txt", np.ascontiguousarray(d["Wt"].T))           # M_fea lines of P values
    vals, _ = _run(bench_exe, "--adj", tmp_path / "adj.txt", "--fea", tmp_path / "fea.txt", "--weights",
                   tmp_path / "w.txt", "--p", 21, "--exact", "--spmm-block", 4, "--rows", "0,31")
    assert len(vals) == 42
    assert_prints_csim_log(vals)            # 40 values to the printed digit, the two known entries one ulp off


def test_cpp_host_default_mode_and_dense_features(bench_exe, tmp_path, oracle):
    """fp16 storage / fp32 accumulation and fp32, sparse and dense features, ReLU, timing loop."""
    m = load("mol")
    _write_csr(tmp_path / "adj.txt", m["adj"])
    _write_csr(tmp_path / "fea.txt", m["fea"])
    _write_rows(tmp_path / "fea_dense.txt", m["fea_dense"])
    _write_rows(tmp_path / "w.txt", np.ascontiguousarray(m["Wt"].T))
    P = m["Wt"].shape[0]
    want = oracle.layer_f64(0, 1, m["adj"], m["fea"], m["Wt"], h_round=0)
    for extra, tol in ((["--dtype", "f32"], 1e-5), ([], 1e-2)):
        for fea in (["--gemm-mode", 0, "--fea", tmp_path / "fea.txt"], ["--gemm-mode", 1, "--fea", tmp_path / "fea_dense.txt"]):
            vals, text = _run(bench_exe, "--adj", tmp_path / "adj.txt", "--weights", tmp_path / "w.txt", "--p", P, "--relu", 1,
                              "--rows", "0,5,2272", "--time", 3, *fea, *extra)
            assert "layer time:" in text and len(vals) == 3 * P
            for (i, j), v in vals.items():
                assert abs(float(v) - want[i, j]) <= tol * max(1.0, abs(want[i, j])) + 2e-3 * (tol > 1e-3)


def test_cpp_host_reports_bad_input(bench_exe, tmp_path):
    (tmp_path / "bad.txt").write_text("0, 1\n0\n")
    out = subprocess.run([bench_exe, "--adj", str(tmp_path / "bad.txt"), "--fea", str(tmp_path / "bad.txt"), "--weights",
                          str(tmp_path / "bad.txt"), "--p", "2"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "expected three lines" in out.stderr
