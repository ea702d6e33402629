"""`python bench.py --gpus N` must start N ranks itself: launcher plumbing on the CPU with gloo and a
stub step -- rendezvous on 127.0.0.1, RANK / LOCAL_RANK / WORLD_SIZE per child, rank 0's ONE JSON line relayed, n_gpus = N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_bench_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-cpu"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "env-sharded x2" and d["steps"] == 5 and d["warmup"] == 1
    assert d["collective"] == {"backend": "gloo", "world_size": 2, "rank_sum_check": True} and d["cpu_baseline"] == "N=1 only"


def test_a_dying_rank_takes_the_job_down():
    """the launcher used to wait for its children one after the other -- rank 1 dying left rank 0 in the
    rendezvous for ever.  Now every child is polled, the survivor is terminated and the failing code is returned."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-cpu"], env={"BEZ_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_single_rank_unchanged():
    r = _run(["--steps", "5", "--warmup", "1", "--stub-cpu"])
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1


def test_bench_refuses_world_size_mismatch():
    r = _run(["--gpus", "2", "--steps", "1", "--stub-cpu"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_under_torch_distributed_run():
    """the driver's own launch form for N > 1"""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-cpu"],
                       env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2
