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
        # aggregate-first order over the same exchanges: the rows of X travel, act((A.X).W) follows
        want_sw = ops.layer_forward(A, X, Wt, relu=True, order="aggregate_first")
        s1 = D.layer_allgather(backend, ops.Csr(rp, ci, va, n), X[lo:hi].contiguous(), Wt, True, bounds, aggregate_first=True)
        s2 = D.layer_halo(backend, ops.Csr(rp, plan.col_compact, va, plan.n_table), X[lo:hi].contiguous(), Wt, True, plan,
                          aggregate_first=True)
        assert torch.equal(s1, want_sw[lo:hi]) and torch.equal(s2, want_sw[lo:hi])
        s3 = D.layer_halo_overlap(backend, ops.Csr(*own, plan.n_own), ops.Csr(*far, max(1, sum(plan.recv_counts))),
                                  X[lo:hi].contiguous(), Wt, True, plan, aggregate_first=True)
        assert torch.allclose(s3.float(), want_sw[lo:hi].float(), rtol=2e-3, atol=2e-3)
        # GAT shards the same way (row-local softmax): own row r is row r of the compact table
        att = ((torch.rand(2 * p, generator=g, device=dev) - 0.5) / 2).half()
        want_gat = ops.layer_forward(A, X, Wt, relu=True, gat_attention=att)
        d4 = D.layer_halo(backend, ops.Csr(rp, plan.col_compact, va, plan.n_table), X[lo:hi].contiguous(), Wt, True, plan,
                          attention=att)
        assert torch.equal(d4, want_gat[lo:hi])
        # rows without a live edge (every value of the row <= 0, as a coarse quantiser leaves them): one GPU gives such a
        # row the mean of ALL rows of Wh (SG.py:638-641); the partitioned layer reduces the column sums across ranks
        # and gives the same -- on both ranks, for dead rows of either partition
        dead = torch.zeros(n, dtype=torch.bool, device=dev)
        dead[torch.arange(3, n, 97, device=dev)] = True
        rows_all = torch.repeat_interleave(torch.arange(n, device=dev), A.rowptr.diff().long())
        val2 = torch.where(dead[rows_all], torch.zeros_like(A.val), A.val)
        A2 = ops.Csr(A.rowptr, A.col, val2, n)
        assert A2.has_dead_rows
        want2, _, S2 = ops.layer_forward(A2, X, Wt, relu=True, gat_attention=att, want_edge_outputs=True)
        rp2, ci2, va2 = D.slice_rows(A2.rowptr, A2.col, A2.val, lo, hi)
        A2_loc = ops.Csr(rp2, plan.col_compact, va2, plan.n_table)
        # the SAME plan with another adjacency: the dead-row decision belongs to the adjacency's values, not to the plan
        # (the adjacency of d4 above has no such row and decided "no fill")
        d5 = D.layer_halo(backend, A2_loc, X[lo:hi].contiguous(), Wt, True, plan, attention=att)
        assert D.any_rank_has_dead_rows(A2_loc) is True and bool(dead[lo:hi].any())
        assert torch.allclose(d5.float(), want2[lo:hi].float(), rtol=2e-3, atol=2e-3)
        want2_rows = ops.layer_forward(A2, X, Wt, relu=True, gat_attention=att)       # (the same form of the aggregate: no side outputs)
        assert torch.equal(d5[~dead[lo:hi]], want2_rows[lo:hi][~dead[lo:hi]])       # live rows: the same bits as on one GPU
        H_all = ops.xw_dense(X, Wt).float()
        assert torch.allclose(d5[dead[lo:hi]].float(), torch.relu(H_all.mean(0)).expand(int(dead[lo:hi].sum()), p), rtol=2e-3, atol=2e-3)
        d6 = D.layer_halo(backend, A2_loc, X[lo:hi].contiguous(), Wt, True, plan, attention=att, fill_dead_rows=False)
        assert not d6[dead[lo:hi]].any()                                             # the sparse answer stays available
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_rank():
    port = 29700 + os.getpid() % 1000
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
        assert all(os.path.exists(os.path.join(tmp, f"ok{r}")) for r in range(2))


@pytest.mark.parametrize("extra,expect", [([], "overlapped with the aggregation"),
                                          (["--exchange", "halo"], "gloo all-to-all of halo rows"),
                                          (["--cut", "1.0"], "gloo all-gather")])
def test_bench_multi_rank_path_rehearsal(extra, expect):
    """bench.py as the driver launches it for N > 1, on the small workload, 2 ranks on one GPU over gloo:
    the default partitioned graph (halo exchange) and the no-locality case (all-gather)."""
    env = dict(os.environ, SGX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(29900 + os.getpid() % 80 + len(expect) % 17),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--workload", "small"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert expect in rec["config"]["exchange"] and "RCCL" not in rec["config"]["exchange"]     # labelled by the backend that ran
    # the line proves what carried the exchange: backend, the world size the process group reports, one device
    # identity per rank, and whether the overlapped exchange had to fall back
    assert rec["backend"] == "gloo" and rec["world_size"] == 2 and rec["exchange_fallback"] is False
    assert len(rec["devices"]) == 2 and all("cuda:" in d for d in rec["devices"])
    # both exchanges in the one invocation: the configured one alone, and the no-locality all-gather of H in the
    # library's collective and as one batch of point-to-point transfers
    ag = rec["exchange_allgather"]
    assert ag["all_gather_into_tensor_ms"] > 0 and ag["point_to_point_batch_ms"] > 0
    assert ag["bytes_per_link_per_layer"] == rec["config"]["nodes_per_gpu"] * rec["config"]["hidden"] * 2
    assert abs(ag["all_gather_into_tensor_GBps_per_link"] - ag["bytes_per_link_per_layer"] / ag["all_gather_into_tensor_ms"] / 1e6) < 1e-6 * ag["all_gather_into_tensor_GBps_per_link"] + 1e-9
    # the exchange on its own (outside the timed region): rows, bytes per link, time
    x = rec["exchange"]
    assert x["ms_alone_per_layer"] > 0 and x["rows_received_per_rank_per_layer"] > 0
    assert x["bytes_received_per_rank_per_layer"] == x["rows_received_per_rank_per_layer"] * rec["config"]["hidden"] * 2
    assert 0 < x["max_bytes_per_link_per_layer"] <= x["bytes_received_per_rank_per_layer"] * 2


def test_bench_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no torch.distributed.run in front, the shape of the driver's N = 1 command): the
    parent starts the two ranks as a child process and relays rank 0's line -- ONE line on stdout, the same schema."""
    import json
    env = dict(os.environ, SGX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with tempfile.TemporaryDirectory() as tmp:
        env["SGX_BENCH_PARENT_REPORT"] = os.path.join(tmp, "parent.json")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                              "--workload", "small"], env=env, capture_output=True, text=True, timeout=900)
        rep = json.load(open(env["SGX_BENCH_PARENT_REPORT"]))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size"] == 2 and rec["backend"] == "gloo" and rec["value"] > 0
    assert len(rec["devices"]) == 2 and "exchange" in rec and "exchange_allgather" in rec and "roofline" in rec
    assert rep["torch_imported"] is False and rep["rc"] == 0 and rep["cmd"][1:3] == ["-m", "torch.distributed.run"]


def test_bench_falls_back_to_the_one_pass_halo_exchange():
    """If the overlapped exchange raises in warm-up, all ranks agree to run the one-pass halo form."""
    import json
    env = dict(os.environ, SGX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SGX_BENCH_TEST_FAIL_OVERLAP="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(29300 + os.getpid() % 90),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", "small"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert "overlapped" not in rec["config"]["exchange"] and "all-to-all of halo rows" in rec["config"]["exchange"]
    assert "injected failure" in out.stderr and rec["value"] > 0
    assert rec["exchange_fallback"] is True                     # in the line, not only on stderr


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
    assert rl["traffic"] is None or rl["traffic_source"]          # never a byte count without saying where it is from
    assert rl["kernel"].startswith("spmm_kernel<f16,8,8>") and rec["backend"] is None and rec["world_size"] == 1
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "edges/s"
    assert rec["value"] > cb["value"]
    # the reference's own CPU formulation (torch.mm + torch.spmm) beside the port, at one thread and at the granted cores
    rf = cb["reference_formulation"]
    assert "torch.sparse.mm" in rf["formulation"] and rf["threads_1"]["threads"] == 1
    assert rf["threads_1"]["edges_per_s"] > 0 and rf["threads_granted"]["edges_per_s"] > 0


_RCCL_SELF = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from sgracex1_amd import dist as D, graphs, ops
n, k, f_in, p = 5000, 700, 80, 64
A_own = graphs.uniform_graph(n, 40000, seed=11, device=dev)
g = torch.Generator(device=dev); g.manual_seed(12)
X = torch.rand((n, f_in), generator=g, device=dev).half()
Wt = ((torch.rand((p, f_in), generator=g, device=dev) - 0.5) / 4).half()
# a one-rank world that sends k of its own rows to itself: the RCCL calls, their stream ordering and
# the asynchronous wait are the ones a multi-GPU node runs
send_rows = torch.randperm(n, generator=g, device=dev)[:k].sort().values
far = graphs.uniform_graph(n, 9000, seed=13, device=dev)
far_col = (far.col.long() % k).int()
A_far = ops.Csr(far.rowptr, far_col, far.val, k)
plan = D.HaloPlan(bounds=[0, n], rank=0, col_compact=A_own.col, send_rows=send_rows, send_counts=[k], recv_counts=[k],
                  n_own=n, send_rows32=send_rows.int())
backend = D.hip_backend()
H = ops.xw_dense(X, Wt)
part = ops.spmm_acc(A_own, H, partial_out=True)
want = ops.spmm_acc(A_far, H.index_select(0, send_rows).contiguous(), relu=True, acc_in=part)
for _ in range(3):                                          # repeated: a missing wait would show as stale halo rows
    got = D.layer_halo_overlap(backend, A_own, A_far, X, Wt, True, plan)
    assert torch.equal(got, want)
# the one-pass halo form over the compact table [own rows | received rows]
rp = A_own.rowptr.long() + A_far.rowptr.long()
own_deg, far_deg = A_own.rowptr.diff().long(), A_far.rowptr.diff().long()
rows_o = torch.repeat_interleave(torch.arange(n, device=dev), own_deg)
rows_f = torch.repeat_interleave(torch.arange(n, device=dev), far_deg)
rows = torch.cat([rows_o, rows_f]); cols = torch.cat([A_own.col.long(), A_far.col.long() + n])
vals = torch.cat([A_own.val, A_far.val])
order = torch.argsort(rows * (n + k) + cols)
A_all = ops.Csr(rp.int(), cols[order].int(), vals[order], n + k)
plan2 = D.HaloPlan(bounds=[0, n], rank=0, col_compact=A_all.col, send_rows=send_rows, send_counts=[k], recv_counts=[k], n_own=n,
                   send_rows32=send_rows.int())
assert torch.equal(ops.pack_rows(H, send_rows.int()), H.index_select(0, send_rows))        # the pack kernel itself
got2 = D.layer_halo(backend, A_all, X, Wt, True, plan2)
table = torch.cat([H, H.index_select(0, send_rows)])
assert torch.equal(got2, ops.spmm(A_all, table, relu=True))
assert torch.allclose(got2.float(), want.float(), rtol=2e-3, atol=2e-3)
# the index exchange of build_halo_plan on RCCL (int64 device tensors, equal and empty splits)
auto = D.build_halo_plan(A_own.col, [0, n], 0)
assert auto.n_own == n and auto.recv_counts == [0] and auto.send_counts == [0]
assert torch.equal(auto.col_compact, A_own.col)
got3 = D.layer_allgather(backend, A_own, X, Wt, True, [0, n])
assert torch.equal(got3, ops.spmm(A_own, H, relu=True))
assert torch.equal(D.layer_allgather(backend, A_own, X, Wt, True, [0, n], direct=True), got3)
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl-self-ok")
'''


def test_exchanges_over_rccl_single_rank():
    """The three exchanges on the real RCCL backend (`nccl`), in a one-rank world that sends rows to
    itself -- the box has one GPU, RCCL refuses two ranks on it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _RCCL_SELF, ROOT, str(29400 + os.getpid() % 500)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl-self-ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
