"""`python bench.py --gpus N` (N > 1) starts its own ranks and never ends without a line on stdout.

Runs without a GPU: the ranks cannot compute here (the product path has no CPU form), so what is checked is the parent --
the child command it builds, that it stays clear of torch and the GPU, and that every way a rank can end (an exception, a
kill from outside, the time limit) gives ONE parsable {"error": ...} line and a non-zero exit code.  The path that succeeds
is tests/test_gpu_dist.py::test_bench_starts_its_own_ranks (two ranks on the box's GPU over gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(tmp_path, extra_env, args=("--gpus", "2", "--workload", "small", "--steps", "2", "--warmup", "1"), timeout=300):
    report = tmp_path / "parent.json"
    env = dict(os.environ, SGX_BENCH_PARENT_REPORT=str(report), **extra_env)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=timeout, cwd=str(tmp_path))
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    return out, lines, json.load(open(report))


def test_a_failing_rank_gives_one_error_line_and_the_parent_never_touches_torch(tmp_path):
    out, lines, rep = _run(tmp_path, {"SGX_BENCH_TEST_FAIL_RANK": "1"})
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout                      # ONE line, whatever the ranks printed
    rec = json.loads(lines[0])
    assert "error" in rec and rec["n_gpus"] == 2 and rec["rc"] != 0 and rec["timed_out"] is False
    assert any(e.get("rank") == 1 and "injected failure" in e["error"] for e in rec["rank_errors"])
    assert "injected failure" in out.stderr                 # the rank's traceback went through
    # the parent: a child process (not an exec), the driver's own launch line, no torch -- hence no GPU -- in this process
    assert rep["torch_imported"] is False and rep["rc"] == rec["rc"]
    cmd = rep["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(BENCH) + 1:] == ["--gpus", "2", "--workload", "small", "--steps", "2", "--warmup", "1"]


def test_a_rank_killed_from_outside_still_gives_a_line(tmp_path):
    """SIGKILL leaves no traceback and no line of the rank's own (what an out-of-memory kill or a process guard does):
    the parent reports the launcher's return code and the tail of its stderr."""
    out, lines, rep = _run(tmp_path, {"SGX_BENCH_TEST_DIE_RANK": "0"})
    assert out.returncode != 0 and len(lines) == 1
    rec = json.loads(lines[0])
    assert "error" in rec and rec["result_lines_seen"] == 0
    assert any("Signal 9" in ln or "SIGKILL" in ln or "exitcode  : -9" in ln for ln in rec["stderr_tail"]), rec["stderr_tail"]


def test_the_time_limit_ends_the_ranks_and_says_so(tmp_path):
    out, lines, rep = _run(tmp_path, {"SGX_BENCH_TEST_HANG_RANK": "all"},
                           args=("--gpus", "2", "--workload", "small", "--rank-timeout", "20"))
    assert out.returncode != 0 and len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["timed_out"] is True and "rank-timeout" in rec["error"] and rep["timed_out"] is True
    # nothing of the job is left behind: the child ran in a session of its own and that process group is empty now
    groups = subprocess.run(["ps", "-eo", "pgid="], capture_output=True, text=True).stdout.split()
    assert str(rep["child_pid"]) not in groups


def test_one_gpu_failure_is_a_line_too(tmp_path):
    """N = 1 runs in the calling process: an exception still ends in a traceback on stderr + one error line + rc 1."""
    env = dict(os.environ, SGX_BENCH_TEST_FAIL_RANK="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, BENCH, "--workload", "small"], env=env, capture_output=True, text=True, timeout=120)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert out.returncode == 1 and len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["rank"] == 0 and "injected failure" in rec["error"] and "Traceback" in out.stderr
