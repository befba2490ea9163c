#!/usr/bin/env python3
"""Headline benchmark: edges aggregated per second (+ achieved HBM GB/s) of a 2-layer GCN forward.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s100m|s100m-rmat|reddit|cora]

A "step" is one 2-layer forward over one synthetic graph already resident in HBM:
    layer 1:  D1 = relu(A . (X . W1))    X sparse CSR (gemm_mode 0), F_in -> hidden
    layer 2:  D2 =      A . (D1 . W2)    X dense      (gemm_mode 1), hidden -> hidden
through sgx_layer_forward (the drop-in for mmult_top).  Default workload = the configuration
BASELINE.json's target is quoted on: S-100M -- N = 2^22 nodes, 100 M uniformly random directed
edges (+ self loops, coalesced), GCN symmetric normalisation, fp16, hidden = 64, Cora-like
sparse input features (F_in = 1433, density 1.27 %).  Output: ONE JSON line on rank 0.

N > 1 (one rank per GPU; `python bench.py --gpus N` starts the ranks itself as a child `torch.distributed.run`, and a
rank started by the driver's own `python -m torch.distributed.run ... bench.py --gpus N` finds WORLD_SIZE set and just
runs): weak scaling -- every rank owns a
partition of 2^22 rows / 100 M edges of a graph with N * 2^22 nodes; --cut (default 0.1) of a
partition's edges leave it and land on the boundary nodes (--boundary, default 0.2 of the rows) of
the other partitions, the rest stay inside it -- the shape a graph partitioner leaves behind (N = 1
is exactly the single-GPU graph).  Per layer the rows of H = X.W that other partitions reference are exchanged
over xGMI (RCCL all-to-all of halo rows; --exchange allgather / --cut 1.0 = every row) and
aggregated locally (sgracex1_amd/dist.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (log2 N or N, edges, F_in, hidden, generator, feature density)
    "s100m": dict(n=1 << 22, edges=100_000_000, f_in=1433, hidden=64, gen="uniform", x_density=0.0127),
    "s100m-rmat": dict(n=1 << 22, edges=100_000_000, f_in=1433, hidden=64, gen="rmat", x_density=0.0127),
    "reddit": dict(n=232_965, edges=114_600_000, f_in=602, hidden=128, gen="uniform", x_density=None),
    "small": dict(n=1 << 16, edges=1_000_000, f_in=256, hidden=64, gen="uniform", x_density=0.05),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="s100m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact", action="store_true",
                    help="N = 1: run both layers in SGX_ACC_REF_HALF with SPMM_BLOCK 4 -- the reference's half arithmetic, "
                         "bit for bit (the setting that reproduces its csim log)")
    ap.add_argument("--cpu-sample-frac", type=float, default=1.0)
    ap.add_argument("--cut", type=float, default=0.1,
                    help="N > 1: share of a partition's edges whose column is drawn from the whole graph "
                         "(1.0 = no locality at all, every H row is a halo row)")
    ap.add_argument("--boundary", type=float, default=0.2,
                    help="N > 1: share of a partition's nodes that edges from other partitions may point at "
                         "(1.0 = any node: every remote row ends up in somebody's halo)")
    ap.add_argument("--exchange", choices=("halo-overlap", "halo", "allgather"), default=None,
                    help="N > 1: rows of H exchanged per layer (default: halo-overlap = halo rows travel while the "
                         "own-partition edges are aggregated; allgather when --cut >= 0.5)")
    ap.add_argument("--no-rmat-leg", action="store_true",
                    help="N = 1, workload s100m: skip the power-law (R-MAT) variant of the aggregation launch that is timed "
                         "after the step and reported as roofline_rmat (outside `value`)")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 started without torch.distributed.run: seconds the ranks get before the parent ends "
                         "them and prints an error line")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="rocprofv3 --pmc result for the dominant kernel (HBM bytes per launch)")
    return ap.parse_args()


def make_inputs(torch, graphs, ops, wl, rank, world, device, cut=1.0, boundary=1.0):
    """Synthetic graph + features + weights of the stated shape, generated on the device."""
    n, hidden, f_in = wl["n"], wl["hidden"], wl["f_in"]
    seed = 12345 + rank
    n_global = n * world
    if world == 1:
        if wl["gen"] == "rmat":
            A = graphs.rmat_graph(n.bit_length() - 1, wl["edges"], seed=seed, device=device)
        else:
            A = graphs.uniform_graph(n, wl["edges"], seed=seed, device=device)
    else:
        # this rank's row block of a graph over n_global nodes: rows local, columns global
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        row = torch.randint(0, n, (wl["edges"],), generator=g, device=device, dtype=torch.int64)
        # a partitioned graph: an edge leaves its row's partition with probability `cut`; it then lands in
        # another partition, on one of that partition's boundary nodes (its first `boundary` share of rows --
        # a partitioner leaves most nodes interior); otherwise the edge stays inside the partition
        n_bnd = max(1, int(n * boundary))
        peer = torch.randint(0, world - 1, (wl["edges"],), generator=g, device=device, dtype=torch.int64)
        peer = peer + (peer >= rank).to(torch.int64)                    # any partition but the own one
        col_far = peer * n + torch.randint(0, n_bnd, (wl["edges"],), generator=g, device=device, dtype=torch.int64)
        col_own = torch.randint(0, n, (wl["edges"],), generator=g, device=device, dtype=torch.int64) + rank * n
        leaves = torch.rand(wl["edges"], generator=g, device=device) < cut
        col = torch.where(leaves, col_far, col_own)
        del col_far, col_own, leaves, peer
        loops = torch.arange(n, device=device, dtype=torch.int64)
        row = torch.cat([row, loops])
        col = torch.cat([col, loops + rank * n])
        key = torch.unique(row * n_global + col)
        row = torch.div(key, n_global, rounding_mode="floor")
        col = key - row * n_global
        counts = torch.bincount(row, minlength=n)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        val = (1.0 / counts.to(torch.float32))[row]            # row normalisation: needs no remote degrees
        A = ops.Csr(rowptr.to(torch.int32), col.to(torch.int32), val.half(), n_global)
        del key, row, col
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1000)
    if wl["x_density"] is not None:
        nnz_x = int(n * f_in * wl["x_density"])
        xr = torch.randint(0, n, (nnz_x,), generator=g, device=device, dtype=torch.int64)
        xc = torch.randint(0, f_in, (nnz_x,), generator=g, device=device, dtype=torch.int64)
        key = torch.unique(xr * f_in + xc)
        xr = torch.div(key, f_in, rounding_mode="floor")
        xc = key - xr * f_in
        xp = torch.zeros(n + 1, dtype=torch.int64, device=device)
        torch.cumsum(torch.bincount(xr, minlength=n), 0, out=xp[1:])
        X = ops.Csr(xp.to(torch.int32), xc.to(torch.int32), torch.ones(key.numel(), device=device).half(), f_in)
        del key, xr, xc
    else:
        X = torch.rand((n, f_in), generator=g, device=device).half()
    bound = 1.0 / hidden ** 0.5                                   # MOL cell 17 reset_parameters
    W1t = ((torch.rand((hidden, f_in), generator=g, device=device) * 2 - 1) * bound).half()
    W2t = ((torch.rand((hidden, hidden), generator=g, device=device) * 2 - 1) * bound).half()
    return A, X, W1t, W2t


def cpu_port_aggregate(rowptr, col, val, H, threads):
    """Part of the cpu_baseline leg (the only place outside tests/ and smoke() that runs the oracle): one A.H
    pass of the oracle's C loops over numpy CSR arrays on `threads` threads; returns a callable for timing
    (tools/cpu_paths.py puts it beside torch.sparse.mm and scipy)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    rows, hidden = rowptr.shape[0] - 1, H.shape[1]
    out = np.zeros((rows, hidden), np.float32)
    blocks = [(rows * t // threads, rows * (t + 1) // threads) for t in range(threads)]

    def run():
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda b: O.spmm_f32_into(0, rowptr, col, val, H, out, b[0], b[1], hidden), blocks))
    return run


def cpu_baseline(torch, ops, A, X, W1t, W2t, frac):
    """The oracle's plain C loops on this box's host cores, on a bounded row sample of the
    same workload: rows [0, frac*N) of all four stages, full H tables (baseline only)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    n = A.n_rows
    rows = max(64, int(n * frac))
    threads = max(1, min(os.cpu_count() or 1, 32))
    hidden = W1t.shape[0]
    # device-side inputs the sample needs, as fp32 on the host
    if isinstance(X, ops.Csr):
        W1 = ops.transpose(W1t).float().cpu().numpy()                              # [F_in, hidden]
        H1 = ops.spmm(X, ops.transpose(W1t), relu=False, use_plan=False).float().cpu().numpy()
        xe = int(X.rowptr[rows])
        xrp, xci, xva = (X.rowptr[:rows + 1].cpu().numpy(), X.col[:xe].cpu().numpy(),
                         X.val[:xe].float().cpu().numpy())
    else:
        W1 = np.ascontiguousarray(W1t.float().cpu().numpy().T)
        H1 = ops.xw_dense(X, W1t).contiguous().float().cpu().numpy()
        Xs = X[:rows].float().cpu().numpy()
    D1 = ops.spmm(A, torch.as_tensor(H1, device=W1t.device).half(), relu=True)
    H2 = ops.xw_dense(D1, W2t).contiguous().float().cpu().numpy()
    D1s = D1[:rows].float().cpu().numpy()
    W2 = np.ascontiguousarray(W2t.float().cpu().numpy().T)
    ae = int(A.rowptr[rows])
    arp, aci, ava = A.rowptr[:rows + 1].cpu().numpy(), A.col[:ae].cpu().numpy(), A.val[:ae].float().cpu().numpy()
    out = np.zeros((rows, hidden), np.float32)
    blocks = [(rows * t // threads, rows * (t + 1) // threads) for t in range(threads)]

    def run(fn):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda b: fn(*b), blocks))
        return time.perf_counter() - t0

    t = 0.0
    if isinstance(X, ops.Csr):
        t += run(lambda lo, hi: O.spmm_f32_into(0, xrp, xci, xva, W1, out, lo, hi, hidden))
    else:
        t += run(lambda lo, hi: O.xw_dense_f32_into(Xs, W1, out, lo, hi))
    t += run(lambda lo, hi: O.spmm_f32_into(1, arp, aci, ava, H1, out, lo, hi, hidden))
    t += run(lambda lo, hi: O.xw_dense_f32_into(D1s, W2, out, lo, hi))
    t += run(lambda lo, hi: O.spmm_f32_into(0, arp, aci, ava, H2, out, lo, hi, hidden))
    return {"value": 2.0 * ae / t, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"rows [0,{rows}) of both layers ({ae} edges per layer), all four stages, full H tables, "
                      f"fp32, {t:.2f} s of wall time on {threads} threads"}


def cpu_reference_formulation(torch, ops, A, X, W1t, W2t, frac, budget_s=12.0):
    """The reference's own CPU path for the same layers -- `support = torch.mm(input, weight)`, `output =
    torch.spmm(adj, support)` (GNN_arc.pdf p.13 Listing 1.3; Graph_Classification.ipynb cell 17, acc == 0) -- in fp32
    on this box's host cores: rows [0, frac N) of both layers against the full H tables, at one thread and at the
    granted core count.  Baseline only; repetitions stop at `budget_s` seconds per thread setting."""
    rows = max(64, int(A.n_rows * frac))
    hidden = W1t.shape[0]
    ae = int(A.rowptr[rows])
    A_s = torch.sparse_csr_tensor(A.rowptr[:rows + 1].cpu().long(), A.col[:ae].cpu().long(), A.val[:ae].float().cpu(),
                                  size=(rows, A.n_cols))
    W1, W2 = W1t.float().cpu().t().contiguous(), W2t.float().cpu().t().contiguous()
    if isinstance(X, ops.Csr):
        xe = int(X.rowptr[rows])
        X_s = torch.sparse_csr_tensor(X.rowptr[:rows + 1].cpu().long(), X.col[:xe].cpu().long(), X.val[:xe].float().cpu(),
                                      size=(rows, X.n_cols))
        H1 = ops.spmm(X, ops.transpose(W1t), relu=False, use_plan=False).float().cpu()
    else:
        X_s = X[:rows].float().cpu()
        H1 = ops.xw_dense(X, W1t).contiguous().float().cpu()
    D1 = ops.spmm(A, H1.to(W1t.device).half(), relu=True)
    H2 = ops.xw_dense(D1, W2t).contiguous().float().cpu()
    D1_s = D1[:rows].float().cpu()

    def forward():
        s1 = torch.sparse.mm(X_s, W1) if X_s.layout != torch.strided else torch.mm(X_s, W1)     # support of layer 1 (sample rows)
        o1 = torch.relu(torch.sparse.mm(A_s, H1))
        s2 = torch.mm(D1_s, W2)
        o2 = torch.sparse.mm(A_s, H2)
        return s1, o1, s2, o2

    out = {"formulation": "torch.sparse.mm(adj_csr, support), support = torch.(sparse.)mm(input, weight), fp32 "
                          "(GNN_arc.pdf Listing 1.3; Graph_Classification.ipynb cell 17, acc == 0)",
           "sample": f"rows [0,{rows}) of both layers ({ae} edges per layer), all four products, full H tables"}
    before = torch.get_num_threads()
    granted = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for label, k in (("threads_1", 1), ("threads_granted", max(1, min(granted, 64)))):
        torch.set_num_threads(k)
        forward()                                                     # warm-up
        ts, t_all = [], time.perf_counter()
        while len(ts) < 5 and (time.perf_counter() - t_all < budget_s or not ts):
            t0 = time.perf_counter()
            forward()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[label] = {"threads": k, "edges_per_s": 2.0 * ae / ts[len(ts) // 2], "runs": len(ts), "median_s": ts[len(ts) // 2]}
    torch.set_num_threads(before)
    return out


def rmat_leg(torch, graphs, ops, Event, wl, device, stream, launches, warmup):
    """The power-law variant (p) of S-100M (R-MAT .57/.19/.19/.05, seed 12345), aggregation launch only:
    D = relu(A . H) at the same N / edges / hidden, timed by HIP events on the launch stream AFTER the step's timed region
    and reported as `roofline_rmat` -- never part of `value`.  The measured-bytes fraction quotes the committed counter
    passes of this workload (counters need their own rocprofv3 --pmc runs), with its source named."""
    n, hidden = wl["n"], wl["hidden"]
    A = graphs.rmat_graph(n.bit_length() - 1, wl["edges"], seed=12345, device=device)
    A.plan
    g = torch.Generator(device=device)
    g.manual_seed(4321)
    H = torch.rand((n, hidden), generator=g, device=device).half()
    D = torch.empty((n, hidden), dtype=torch.float16, device=device)
    for _ in range(warmup):
        ops.spmm(A, H, relu=True, out=D)
    pairs = [(Event(), Event()) for _ in range(launches)]
    for b, e in pairs:
        b.record(stream)
        ops.spmm(A, H, relu=True, out=D)
        e.record(stream)
    torch.cuda.synchronize()
    ms = sorted(b.elapsed_ms(e) for b, e in pairs)
    avg = sum(ms) / len(ms)
    nnz = A.nnz
    b_alg = nnz * (4 + 2 + hidden * 2) + (n + 1) * 4 + n * hidden * 2
    deg = (A.rowptr[1:] - A.rowptr[:-1])
    out = {"workload": "s100m-rmat", "generator": "rmat a/b/c/d=.57/.19/.19/.05 seed 12345", "nodes": n, "edges": nnz,
           "hidden": hidden, "max_degree": int(deg.max().item()), "plan_long_rows": A.plan.long_rows,
           "plan_long_threshold": A.plan.long_threshold, "plan_reordered": A.plan.reordered,
           "kernel": "spmm_kernel<f16,8,8> + long-row tasks + spmm_split_finalize_kernel (A.H aggregation, events bracket all of it)",
           "avg_launch_ms": avg, "min_launch_ms": ms[0], "launches_timed": len(ms),
           "algorithmic_bytes_per_launch": b_alg, "achieved": b_alg / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": b_alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "agg_edges_per_s": nnz / (avg * 1e-3),
           "traffic": None, "traffic_source": None, "frac_measured_bytes": None}
    tf_path = os.path.join(ROOT, "profiles", "traffic_s100m-rmat.json")
    if os.path.exists(tf_path):
        try:
            tf = json.load(open(tf_path))
            out["traffic"] = tf.get("hbm_bytes_per_launch")
            out["traffic_source"] = (f"offline rocprofv3 --pmc passes of this workload, {tf.get('source')} (tag {tf.get('tag')}); "
                                     "not collected in this run")
            if out["traffic"]:
                # the binding number where cache hits make the memory side move fewer bytes than the gather model counts
                out["frac_measured_bytes"] = out["traffic"] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS
        except (ValueError, OSError):
            pass
    del A, H, D
    return out


def device_identity(torch, device):
    """What tells one GPU of the node from another in the bench line: host, index, name, PCI bus / uuid where torch has them."""
    import socket
    p = torch.cuda.get_device_properties(device)
    parts = [socket.gethostname(), f"cuda:{device.index}", p.name]
    for attr in ("pci_bus_id", "pci_device_id", "uuid"):
        v = getattr(p, attr, None)
        if v is not None:
            parts.append(f"{attr}={v}")
    return " ".join(str(x) for x in parts)


_WHERE = ["start"]


def note(msg):
    """progress on stderr (SGX_BENCH_VERBOSE=1): where a multi-rank rehearsal is when it is slow or stuck; the last note
    also goes into the error line of a rank that fails"""
    _WHERE[0] = msg
    if os.environ.get("SGX_BENCH_VERBOSE"):
        print(f"[bench {os.environ.get('RANK', '0')} {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def emit(obj):
    """One line on stdout in ONE write (line and newline together): the ranks of a job share the pipe, and two prints that
    interleave between text and newline give the reader one unparsable line."""
    sys.stdout.flush()
    os.write(1, (json.dumps(obj) + "\n").encode())


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: this process only PARENTS the ranks.  It starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a child process (subprocess -- never an
    exec, and nothing here imports torch or touches the GPU), passes the ranks' stderr through, and prints exactly ONE line
    on stdout: rank 0's result line, or -- whatever ended the run (a rank's exception, a rank killed from outside, the
    time limit) -- an {"error": ...} line with the child's return code and the tail of its stderr.  Returns the exit code."""
    import collections
    import signal
    import socket
    import subprocess
    import threading

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on these hosts
    t0 = time.time()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, bufsize=1,
                             start_new_session=True)
    results, rank_errors, tail = [], [], collections.deque(maxlen=40)

    def pump_out():
        for ln in child.stdout:
            ln = ln.rstrip("\n")
            obj = None
            if ln.startswith("{"):
                try:
                    obj = json.loads(ln)
                except ValueError:
                    obj = None
            if isinstance(obj, dict) and "metric" in obj and "value" in obj:
                results.append(ln)
            elif isinstance(obj, dict) and "error" in obj:
                rank_errors.append(obj)
            elif ln:
                print(ln, file=sys.stderr, flush=True)          # library chatter ("[Gloo] Rank 0 is connected ...")

    def pump_err():
        for ln in child.stderr:
            tail.append(ln.rstrip("\n"))
            sys.stderr.write(ln)
            sys.stderr.flush()

    threads = [threading.Thread(target=pump_out, daemon=True), threading.Thread(target=pump_err, daemon=True)]
    for t in threads:
        t.start()
    timed_out = False
    try:
        rc = child.wait(timeout=args.rank_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig, grace in ((signal.SIGTERM, 15), (signal.SIGKILL, 15)):
            try:
                os.killpg(child.pid, sig)                        # the child's own session: the launcher and its ranks, nothing else
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = child.poll() if child.poll() is not None else -9
    for t in threads:
        t.join(timeout=10)
    report = os.environ.get("SGX_BENCH_PARENT_REPORT")
    if report:                                                   # for tests: what the parent did and did not do
        with open(report, "w") as f:
            json.dump({"cmd": cmd, "rc": rc, "timed_out": timed_out, "torch_imported": "torch" in sys.modules, "child_pid": child.pid,
                       "seconds": time.time() - t0}, f)
    if rc == 0 and len(results) == 1:
        print(results[0], flush=True)
        return 0
    why = (f"the ranks did not finish within --rank-timeout {args.rank_timeout:.0f} s and were ended" if timed_out else
           f"torch.distributed.run ended with return code {rc}" if rc != 0 else
           f"the ranks ended with return code 0 but printed {len(results)} result lines")
    print(json.dumps({"error": why, "n_gpus": args.gpus, "rc": rc, "timed_out": timed_out, "rank_errors": rank_errors,
                      "result_lines_seen": len(results), "stderr_tail": list(tail)[-25:], "cmd": cmd}), flush=True)
    return rc if rc not in (0, None) else 1


def main():
    """Entry: parent of the ranks (above), or one rank.  A rank that fails prints its traceback on stderr and ONE
    {"error": ..., "rank": ...} line on stdout and exits non-zero -- an N > 1 run never ends without a line."""
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    try:
        run(args)
    except SystemExit:
        raise
    except BaseException as exc:              # noqa: BLE001 -- KeyboardInterrupt included: say so before going down
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        emit({"error": f"{type(exc).__name__}: {exc}", "rank": int(os.environ.get("RANK", "0")),
              "world_size": int(os.environ.get("WORLD_SIZE", "1")), "where": _WHERE[0]})
        sys.exit(1)


def run(args):
    # test hooks: a rank that raises / a rank that is killed from outside (no traceback, no line of its own)
    if os.environ.get("SGX_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise RuntimeError("injected failure of this rank")
    if os.environ.get("SGX_BENCH_TEST_DIE_RANK") == os.environ.get("RANK", "0"):
        import signal
        os.kill(os.getpid(), signal.SIGKILL)
    if os.environ.get("SGX_BENCH_TEST_HANG_RANK") in (os.environ.get("RANK", "0"), "all"):
        time.sleep(3600)
    import torch
    import torch.distributed as dist
    from sgracex1_amd import dist as sdist
    from sgracex1_amd import graphs, ops
    from sgracex1_amd.hipevents import Event

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                       # under torch.distributed.run the launcher's rank count is the truth
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    if world > 1:
        # "nccl" is RCCL on ROCm; SGX_DIST_BACKEND=gloo rehearses several ranks on one GPU
        backend_name = os.environ.get("SGX_DIST_BACKEND", "nccl")
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend_name)

    note("process group up" if world > 1 else "device set")
    # what the line must prove for N > 1: the backend that actually carried the exchange, the number of ranks IT sees,
    # and one identity per rank (distinct devices = the ranks really sat on different GPUs)
    dist_backend = dist.get_backend() if world > 1 else None
    dist_world = dist.get_world_size() if world > 1 else 1
    devices = [device_identity(torch, device)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    coll = {"nccl": "RCCL"}.get(dist_backend, dist_backend)

    wl = WORKLOADS[args.workload]
    if args.cut >= 0.5:
        args.boundary = 1.0                      # "no locality": any node of any partition
    A, X, W1t, W2t = make_inputs(torch, graphs, ops, wl, rank, world, device, cut=args.cut, boundary=args.boundary)
    exchange = args.exchange or ("allgather" if args.cut >= 0.5 else "halo-overlap")
    n, hidden = wl["n"], wl["hidden"]
    nnz = A.nnz
    note(f"graph ready: {n} rows, {nnz} stored entries")
    A.plan  # build the row schedules outside the timed region (once per graph)
    if isinstance(X, ops.Csr):
        X.plan
    D1 = torch.empty((n, hidden), dtype=torch.float16, device=device)
    D2 = torch.empty((n, hidden), dtype=torch.float16, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    n_ev = 2 * args.steps
    ev = [(Event(), Event()) for _ in range(n_ev)]

    if args.exact and world > 1:
        sys.exit("--exact is a one-GPU measurement (the reference's sequential half arithmetic has no partitioned form here)")
    mode = dict(acc_mode=ops.SGX_ACC_REF_HALF, spmm_block=4) if args.exact else {}
    if world == 1:
        def step(i, timed):
            e1 = (ev[2 * i][0].handle, ev[2 * i][1].handle) if timed else None
            e2 = (ev[2 * i + 1][0].handle, ev[2 * i + 1][1].handle) if timed else None
            ops.layer_forward(A, X, W1t, relu=True, out=D1, agg_events=e1, **mode)
            ops.layer_forward(A, D1, W2t, relu=False, out=D2, agg_events=e2, **mode)
    else:
        backend = sdist.hip_backend()
        bounds = [g * n for g in range(world + 1)]
        halo = None
        overlap = exchange == "halo-overlap"
        agg_nnz = nnz                       # edges of the launch the roofline events bracket
        if exchange in ("halo", "halo-overlap"):
            # once per graph: which rows of H every peer needs, and A's columns renumbered to the
            # compact table [own rows | halo rows by owner]
            halo = sdist.build_halo_plan(A.col, bounds, rank)
            A_run = ops.Csr(A.rowptr, halo.col_compact, A.val, halo.n_table)
            A_run.plan
            table = torch.empty((halo.n_table, hidden), dtype=torch.float16, device=device)
            if overlap:
                own, far = sdist.split_own_halo(A.rowptr, halo.col_compact, A.val, halo.n_own)
                A_own, A_far = ops.Csr(*own, halo.n_own), ops.Csr(*far, max(1, sum(halo.recv_counts)))
                A_own.plan, A_far.plan
                halo_table = torch.empty((max(1, sum(halo.recv_counts)), hidden), dtype=torch.float16, device=device)
                agg_nnz = A_own.nnz
        else:
            A_run = A
            table = torch.empty((n * world, hidden), dtype=torch.float16, device=device)

        def timed_spmm(adj, tab, relu, pair, out):
            if pair is not None:
                pair[0].record(stream)
            ops.spmm(adj, tab, relu=relu, out=out)
            if pair is not None:
                pair[1].record(stream)
            return out

        def step(i, timed):
            p1 = ev[2 * i] if timed else None
            p2 = ev[2 * i + 1] if timed else None
            b1 = sdist.Backend(backend.xw, lambda a, t, r: timed_spmm(a, t, r, p1, D1))
            b2 = sdist.Backend(backend.xw, lambda a, t, r: timed_spmm(a, t, r, p2, D2))
            if overlap:
                if os.environ.get("SGX_BENCH_TEST_FAIL_OVERLAP"):       # test hook: exercise the fallback below
                    raise RuntimeError("injected failure of the overlapped exchange")

                def first_pass(pair):
                    def run(a, t):
                        if pair is not None:
                            pair[0].record(stream)
                        part = ops.spmm_acc(a, t, partial_out=True)
                        if pair is not None:
                            pair[1].record(stream)
                        return part
                    return run
                o1 = sdist.Backend(backend.xw, None, first_pass(p1),
                                   lambda a, t, part, r: ops.spmm_acc(a, t, relu=r, acc_in=part, out=D1))
                o2 = sdist.Backend(backend.xw, None, first_pass(p2),
                                   lambda a, t, part, r: ops.spmm_acc(a, t, relu=r, acc_in=part, out=D2))
                sdist.layer_halo_overlap(o1, A_own, A_far, X, W1t, True, halo, halo_table=halo_table)
                sdist.layer_halo_overlap(o2, A_own, A_far, D1, W2t, False, halo, halo_table=halo_table)
            elif halo is not None:
                sdist.layer_halo(b1, A_run, X, W1t, True, halo, table=table)
                sdist.layer_halo(b2, A_run, D1, W2t, False, halo, table=table)
            else:
                sdist.layer_allgather(b1, A_run, X, W1t, True, bounds, h_global=table)
                sdist.layer_allgather(b2, A_run, D1, W2t, False, bounds, h_global=table)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; for N > 1 it doubles as the trial of the overlapped exchange: should that path raise on this
    # node, every rank drops to the one-pass halo exchange together (same structures) instead of losing the run
    note("structures ready; warm-up")
    failed = 0
    try:
        for _ in range(args.warmup):
            step(0, False)
        torch.cuda.synchronize()
    except Exception as exc:                       # noqa: BLE001 -- any failure of the trial, reported below
        if not (world > 1 and overlap):
            raise
        failed = 1
        print(f"[bench rank {rank}] overlapped halo exchange failed in warm-up: {exc!r}", file=sys.stderr, flush=True)
    exchange_fallback = False
    if world > 1 and overlap:
        flag = torch.tensor([failed], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            exchange_fallback = True
            overlap, exchange, agg_nnz = False, "halo", nnz
            for _ in range(args.warmup):
                step(0, False)
    barrier()
    note("timed steps")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    barrier()
    elapsed = time.perf_counter() - t0
    note(f"timed steps done: {elapsed:.3f} s")
    if world > 1:
        red_dev = device if dist.get_backend() == "nccl" else "cpu"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([nnz], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot)
        total_nnz = int(tot.item())
    else:
        total_nnz = nnz

    # N > 1: the exchange on its own, outside the timed region (exchange time and bytes per link) -- the
    # same collective on the same row lists, 5 rounds between barriers, slowest rank
    exchange_stats = None
    allgather_stats = None
    if world > 1:
        row_bytes = hidden * 2

        def time_alone(fn, rounds=5):
            fn()
            barrier()
            x0 = time.perf_counter()
            for _ in range(rounds):
                fn()
            barrier()
            xt = torch.tensor([(time.perf_counter() - x0) / rounds], dtype=torch.float64, device=red_dev)
            dist.all_reduce(xt, op=dist.ReduceOp.MAX)
            return float(xt.item()) * 1e3

        if halo is not None:
            recv_rows, link_rows = sum(halo.recv_counts), max(max(halo.recv_counts), max(halo.send_counts))
            dest = halo_table[:recv_rows] if overlap else table[halo.n_own:]

            def exchange_once():
                packed = (ops.pack_rows(D1, halo.send_rows32) if halo.send_rows.numel() else D1.new_empty((0, hidden)))
                sdist.all_to_all_rows(dest, packed, halo.recv_counts, halo.send_counts)
            what = f"HIP pack kernel + {coll} all-to-all of the halo rows"
        else:
            recv_rows, link_rows = n * (world - 1), n

            def exchange_once():
                sdist.all_gather_into(table, D1)
            what = f"{coll} all-gather of H"
        note("exchange alone")
        x_ms = time_alone(exchange_once)
        note(f"exchange alone: {x_ms:.2f} ms; all-gather forms")
        exchange_stats = {"what": what, "rows_received_per_rank_per_layer": recv_rows,
                          "bytes_received_per_rank_per_layer": recv_rows * row_bytes,
                          "max_bytes_per_link_per_layer": link_rows * row_bytes, "ms_alone_per_layer": x_ms,
                          "GBps_busiest_link_alone": link_rows * row_bytes / (x_ms * 1e-3) / 1e9,
                          "note": "pack + collective without any aggregation beside it, 5 rounds between barriers, slowest rank's time"}
        # SURVEY 8e asks for both exchanges: the no-locality form (every row of H to every rank) on the same H, in the
        # library's all-gather and as one batch of point-to-point transfers (one per link of the fully connected node)
        full = table if halo is None else torch.empty((n * world, hidden), dtype=torch.float16, device=device)
        ag_ms = time_alone(lambda: sdist.all_gather_into(full, D1), rounds=3)
        note(f"all_gather_into_tensor: {ag_ms:.2f} ms; point-to-point batch")
        p2p_ms = time_alone(lambda: sdist.all_gather_direct(full, D1, bounds, rank), rounds=3)
        note(f"point-to-point batch: {p2p_ms:.2f} ms")
        per_link = n * row_bytes
        allgather_stats = {"what": f"all-gather of H [{n * world} x {hidden}] f16 alone, {world} ranks over {coll}",
                           "bytes_received_per_rank_per_layer": n * (world - 1) * row_bytes, "bytes_per_link_per_layer": per_link,
                           "all_gather_into_tensor_ms": ag_ms, "all_gather_into_tensor_GBps_per_link": per_link / (ag_ms * 1e-3) / 1e9,
                           "point_to_point_batch_ms": p2p_ms, "point_to_point_batch_GBps_per_link": per_link / (p2p_ms * 1e-3) / 1e9,
                           "note": "per-link rate = one rank's block over the time: a direct exchange runs every link at it, "
                                   "a ring moves (ranks - 1) blocks over each link in turn"}
        del full

    # dominant kernel: the A.H aggregation (spmm_kernel), timed by events the launch path
    # recorded on its own stream inside the timed region
    agg_ms = sorted(b.elapsed_ms(e) for b, e in ev)
    agg_avg_ms = sum(agg_ms) / len(agg_ms)
    n_cols = A.n_cols
    es = 2
    if world == 1:
        agg_nnz = nnz
    out_es = 4 if (world > 1 and exchange == "halo-overlap") else es   # the overlapped first pass stores fp32 sums
    b_alg = agg_nnz * (4 + es + hidden * es) + (n + 1) * 4 + n * hidden * out_es   # SURVEY 8d, no-reuse gather model
    b_min = nnz * (4 + es) + (n + 1) * 4 + (n + n_cols) * hidden * es          # compulsory traffic
    achieved = b_alg / (agg_avg_ms * 1e-3) / 1e9
    # HBM-side bytes per launch of the dominant kernel: NOT measured in this run -- counters need their own rocprofv3
    # --pmc passes (tools/prof_round.sh); the committed summary of those passes is quoted with its source
    traffic, traffic_source = None, None
    if world == 1 and os.path.exists(args.traffic_file):
        try:
            tf = json.load(open(args.traffic_file))
            if tf.get("workload") == args.workload:
                traffic = tf.get("hbm_bytes_per_launch")
                traffic_source = f"offline rocprofv3 --pmc passes of this workload, {tf.get('source')} (tag {tf.get('tag')}); not collected in this run"
        except (ValueError, OSError):
            traffic = None
    lanes_per_row = 1
    while lanes_per_row * 8 < hidden and lanes_per_row < 64:
        lanes_per_row *= 2

    # the roof a plain copy reaches on this very device (SURVEY 8d asks for it next to the nominal 8 TB/s)
    copy_gbps = None
    if rank == 0:
        src = torch.empty(1 << 29, dtype=torch.float16, device=device)     # 1 GiB read + 1 GiB written
        dst = torch.empty_like(src)
        from sgracex1_amd._lib import check, lib
        import ctypes

        def copy():
            check(lib.sgx_stream_copy(dst.data_ptr(), src.data_ptr(), src.numel() * 2, ctypes.c_void_p(stream)), "sgx_stream_copy")

        for _ in range(2):
            copy()
        cb, ce = Event(), Event()
        cb.record(stream)
        for _ in range(5):
            copy()
        ce.record(stream)
        copy_gbps = 5 * 2 * src.numel() * 2 / (cb.elapsed_ms(ce) * 1e-3) / 1e9
        del src, dst

    ms_per_step = elapsed / args.steps * 1e3
    line = {
        "metric": "edges aggregated/sec, 2-layer GCN forward",
        "value": 2.0 * total_nnz / (elapsed / args.steps),
        "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (every product and add rounded to f16, 4 partial sums: the reference's arithmetic)" if args.exact
                 else "f16 (f32 accumulate)", "data": "synthetic",
        "config": {"workload": args.workload, "nodes_per_gpu": n, "edges_per_gpu": nnz, "generator": wl["gen"],
                   "f_in": wl["f_in"], "hidden": hidden,
                   "layer1": "gemm_mode=0 sparse X, relu=1" if wl["x_density"] else "gemm_mode=1 dense X, relu=1",
                   "layer2": "gemm_mode=1 dense X, relu=0",
                   "exchange": "none" if world == 1 else
                   (f"{coll} all-to-all of halo rows of H per layer ({sum(halo.recv_counts)} rows received per rank), HIP pack kernel"
                    + (" on a side stream, overlapped with the aggregation of the own-partition edges" if exchange == "halo-overlap" else "")
                    if halo is not None else f"{coll} all-gather of H per layer"),
                   "cut": None if world == 1 else args.cut, "boundary": None if world == 1 else args.boundary},
        "backend": dist_backend, "world_size": dist_world, "devices": devices, "exchange_fallback": exchange_fallback,
        "roofline": {"bound": "hbm", "kernel": ("refhalf_csr_rows_kernel" if args.exact else f"spmm_kernel<f16,8,{lanes_per_row}>")
                     + " (A.H aggregation" + (", own-partition pass)" if (world > 1 and exchange == "halo-overlap") else ")"),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": b_alg, "compulsory_bytes_per_launch": b_min,
                     "avg_launch_ms": agg_avg_ms, "min_launch_ms": agg_ms[0], "launches_timed": len(agg_ms),
                     "agg_edges_per_s": agg_nnz / (agg_avg_ms * 1e-3),
                     "stream_copy_GBps_this_device": copy_gbps},
    }
    if exchange_stats is not None:
        line["exchange"] = exchange_stats
        line["exchange_allgather"] = allgather_stats
    if world == 1 and args.workload == "s100m" and not args.no_rmat_leg and not args.exact:
        note("power-law leg")
        line["roofline_rmat"] = rmat_leg(torch, graphs, ops, Event, wl, device, stream, max(5, min(args.steps, 20)), 3)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(torch, ops, A, X, W1t, W2t, args.cpu_sample_frac)
        line["cpu_baseline"]["reference_formulation"] = cpu_reference_formulation(torch, ops, A, X, W1t, W2t,
                                                                                  min(args.cpu_sample_frac, 0.125))
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
