"""The partitioned layer with the real HIP backend: two ranks share the one GPU of the box and
exchange through gloo (staged through host memory) -- the code path of `bench.py --gpus N`, whose
collective is RCCL on a multi-GPU node."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sgracex1_amd import dist as D, graphs, ops
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        n, f_in, p = 6000, 96, 64
        A = graphs.uniform_graph(n, 60000, seed=7, device=dev)                # same graph on both ranks
        g = torch.Generator(device=dev)
        g.manual_seed(8)
        X = torch.rand((n, f_in), generator=g, device=dev).half()
        Wt = ((torch.rand((p, f_in), generator=g, device=dev) - 0.5) / 4).half()
        want = ops.layer_forward(A, X, Wt, relu=True)
        bounds = D.row_partition(n, world, A.rowptr)
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci, va = D.slice_rows(A.rowptr, A.col, A.val, lo, hi)
        backend = D.hip_backend()
        d1 = D.layer_allgather(backend, ops.Csr(rp, ci, va, n), X[lo:hi].contiguous(), Wt, True, bounds)
        assert torch.equal(d1, want[lo:hi])
        plan = D.build_halo_plan(ci, bounds, rank)
        d2 = D.layer_halo(backend, ops.Csr(rp, plan.col_compact, va, plan.n_table), X[lo:hi].contiguous(), Wt, True, plan)
        assert torch.equal(d2, want[lo:hi])
        own, far = D.split_own_halo(rp, plan.col_compact, va, plan.n_own)
        d3 = D.layer_halo_overlap(backend, ops.Csr(*own, plan.n_own), ops.Csr(*far, max(1, sum(plan.recv_counts))),
                                  X[lo:hi].contiguous(), Wt, True, plan)
        # two passes sum the same terms in another order: equal to fp32 rounding, not bitwise
        assert torch.allclose(d3.float(), want[lo:hi].float(), rtol=2e-3, atol=2e-3)
        assert (d3 == want[lo:hi]).float().mean() > 0.98
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_rank():
    port = 29700 + os.getpid() % 1000
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
        assert all(os.path.exists(os.path.join(tmp, f"ok{r}")) for r in range(2))


@pytest.mark.parametrize("extra,expect", [([], "overlapped with the aggregation"),
                                          (["--exchange", "halo"], "RCCL all-to-all of halo rows"),
                                          (["--cut", "1.0"], "RCCL all-gather")])
def test_bench_multi_rank_path_rehearsal(extra, expect):
    """bench.py as the driver launches it for N > 1, on the small workload, 2 ranks on one GPU over gloo:
    the default partitioned graph (halo exchange) and the no-locality case (all-gather)."""
    env = dict(os.environ, SGX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90 + len(extra)),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--workload", "small"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert expect in rec["config"]["exchange"]


def test_bench_single_gpu_line_has_the_contract_fields():
    """`python bench.py` (N = 1) on the small workload: one JSON line with the driver's fields, the
    roofline object timed on the launch stream, and the CPU baseline from the oracle port."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "4",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 4 and rec["unit"] == "edges/s" and rec["vs_baseline"] is None
    assert rec["config"]["workload"] == "small" and "model" not in rec["config"]
    rl = rec["roofline"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-9 and rl["launches_timed"] == 8
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "edges/s"
    assert rec["value"] > cb["value"]
