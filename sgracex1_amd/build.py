"""Builds sgracex1_amd/csrc/libsgx.so (HIP kernels + the C ABI of include/sgx.h) for gfx950.

    python -m sgracex1_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is built in-tree so that it travels with the
source tree.  There is no other backend and no fallback: without this library the package
cannot run a layer.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(CSRC, "libsgx.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "sgx.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


TESTBENCH_SRC = os.path.join(ROOT, "examples", "sgx_testbench.cpp")
TESTBENCH = os.path.join(ROOT, "examples", "sgx_testbench")


def build_testbench(force=False, verbose=False):
    """examples/sgx_testbench: the C++ host that binds only include/sgx.h (the role of the reference's
    main_float.cpp).  Linked against the in-tree libsgx.so with a relative rpath."""
    build(force=False)
    if (not force and os.path.exists(TESTBENCH)
            and os.path.getmtime(TESTBENCH) >= max(os.path.getmtime(TESTBENCH_SRC), os.path.getmtime(LIB))):
        return TESTBENCH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), TESTBENCH_SRC, "-L", CSRC, "-lsgx",
           "-Wl,-rpath,$ORIGIN/../sgracex1_amd/csrc", "-o", TESTBENCH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TESTBENCH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_testbench(force="--force" in sys.argv, verbose=True))
