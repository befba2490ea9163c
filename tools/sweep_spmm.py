#!/usr/bin/env python3
"""Builds variants of the aggregation kernel (macros of csrc/spmm_csr.hip) and times each on the
S-100M graph in its own process; prints one line per variant.  Run on the GPU box:

    python tools/sweep_spmm.py                 # build all variants (cross-compiles anywhere)
    python tools/sweep_spmm.py --run           # time them (needs the GPU)
"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sgracex1_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "sweep")

VARIANTS = {}
for pieces, mw in itertools.product((1, 2, 3), (1, 3)):
    VARIANTS[f"pieces{pieces}_mw{mw}"] = dict(SGX_SPMM_PIECES=pieces, SGX_SPMM_MINWAVES=mw)


def build():
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "spmm_csr.o"]
    procs = []
    for name, macros in VARIANTS.items():
        obj = os.path.join(OUT, name + ".o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC] + [f"-D{k}={v}" for k, v in macros.items()] + \
              ["-c", os.path.join(CSRC, "spmm_csr.hip"), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
        if len(procs) % 6 == 0:
            for _, _, p in procs[-6:]:
                p.wait()
    for name, obj, p in procs:
        assert p.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, name + ".so"), obj] + others)
        os.remove(obj)


TIMER = r'''
import json, sys, torch
sys.path.insert(0, %r)
from sgracex1_amd import graphs, ops
from sgracex1_amd.hipevents import Event
gen = sys.argv[1]
P = 64
if gen == "rmat":
    A = graphs.rmat_graph(22, 100_000_000)
elif gen == "reddit":
    A = graphs.uniform_graph(232_965, 114_600_000, seed=3); P = 128
else:
    A = graphs.uniform_graph(1 << 22, 100_000_000)
H = torch.rand((A.n_cols, P), device="cuda").half()
D = torch.empty((A.n_rows, P), device="cuda", dtype=torch.float16)
A.plan
for _ in range(3): ops.spmm(A, H, relu=True, out=D)
torch.cuda.synchronize()
s = torch.cuda.current_stream().cuda_stream
ts = []
for _ in range(15):
    b, e = Event(), Event()
    b.record(s); ops.spmm(A, H, relu=True, out=D); e.record(s)
    ts.append(b.elapsed_ms(e))
ts.sort()
print(json.dumps({"min_ms": ts[0], "median_ms": ts[len(ts)//2], "nnz": A.nnz}))
''' % ROOT


def run(gen):
    for name in VARIANTS:
        lib = os.path.join(OUT, name + ".so")
        env = dict(os.environ, SGX_LIB_PATH=lib)
        out = subprocess.run([sys.executable, "-c", TIMER, gen], env=env, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        print(name, line[-1] if line else "FAILED " + out.stderr[-300:], flush=True)


if __name__ == "__main__":
    if "--run" in sys.argv:
        for gen in ("uniform", "rmat", "reddit"):
            print("==", gen, flush=True)
            run(gen)
    else:
        build()
