"""`python bench.py --gpus N` called plainly (as the driver calls it) starts its own ranks before anything touches the GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run_bench(*args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, timeout=timeout)


def test_launcher_starts_the_ranks_and_relays_rank0(lib):
    """--dry-run: the two ranks rendezvous over gloo (what RCCL does on the GPU box), rank 0's line comes back through the
    launcher: world size, every rank present, the read shards contiguous and complete."""
    p = run_bench("--gpus", "2", "--dry-run", "--overlap-reads", "1001")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["rank_sum"] == 1 and d["local_rank_sum"] == 1
    assert d["read_shards"] == [[0, 500], [500, 1001]]


def test_without_a_gpu_every_rank_refuses_and_the_launcher_fails(lib):
    """The real mode on a box without GPUs: the ranks are started (each names its rank), each refuses to run -- there is
    no CPU path to time -- and the launcher exits non-zero without printing a line."""
    import torch
    if torch.cuda.is_available():
        return
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    err = p.stderr.decode()
    assert p.returncode != 0 and p.stdout.decode().strip() == ""
    assert "needs an MI355X" in err and "rank(s) failed" in err
    # mismatch between --gpus and an inherited WORLD_SIZE is an error too, not a silent single-rank run
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, timeout=120)
    assert q.returncode != 0 and b"WORLD_SIZE=3" in q.stderr
