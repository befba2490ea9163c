#!/usr/bin/env python3
"""Variants of the sparse X.W kernel with W in LDS (macros of csrc/xw_sparse_lds.hip), each built into a library of its
own under _variants/ (travels to the GPU box) and timed by tools/xw_sparse_probe.py in its own process, three rounds
interleaved:

    python tools/sweep_xw_sparse.py            # build (cross-compiles anywhere)
    python tools/sweep_xw_sparse.py --run      # time (needs the GPU)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sgracex1_amd", "csrc")
OUT = os.path.join(ROOT, "_variants")

VARIANTS = {
    "pairs1_skip1_d8": dict(SGX_XW_LDS_PAIRS=1, SGX_XW_LDS_SKIP_UNUSED=1, SGX_XW_LDS_DEPTH=8),
    "pairs0_skip1_d8": dict(SGX_XW_LDS_PAIRS=0, SGX_XW_LDS_SKIP_UNUSED=1, SGX_XW_LDS_DEPTH=8),
    "pairs1_skip0_d8": dict(SGX_XW_LDS_PAIRS=1, SGX_XW_LDS_SKIP_UNUSED=0, SGX_XW_LDS_DEPTH=8),
    "pairs1_skip1_d6": dict(SGX_XW_LDS_PAIRS=1, SGX_XW_LDS_SKIP_UNUSED=1, SGX_XW_LDS_DEPTH=6),
    "pairs1_skip1_d10": dict(SGX_XW_LDS_PAIRS=1, SGX_XW_LDS_SKIP_UNUSED=1, SGX_XW_LDS_DEPTH=10),
}


def build():
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "xw_sparse_lds.o"]
    procs = []
    for name, macros in VARIANTS.items():
        obj = os.path.join(OUT, name + ".o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC] + [f"-D{k}={v}" for k, v in macros.items()] + \
              ["-c", os.path.join(CSRC, "xw_sparse_lds.hip"), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, name + ".so"), obj] + others)
        os.remove(obj)


def run():
    res = {}
    for rnd in range(3):
        for name in VARIANTS:
            env = dict(os.environ, SGX_LIB_PATH=os.path.join(OUT, name + ".so"), SGX_PROBE_QUICK="1")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "xw_sparse_probe.py")], env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(name, "FAILED", out.stderr[-400:], flush=True)
                continue
            rec = json.loads(line[-1])
            res.setdefault(name, []).append(min(v[0] for v in rec["ms_lds"]))
            assert rec.get("lds_equal_to_gather", True)
    for name, ts in res.items():
        print(json.dumps({"variant": name, **VARIANTS[name], "ms_lds_min_per_round": ts}), flush=True)


if __name__ == "__main__":
    run() if "--run" in sys.argv else build()
