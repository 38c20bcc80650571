"""N > 1 path on CPU: world_size-2 (and 3) gloo run of the seed-index exchange protocol."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(worker, world):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, worker)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


@pytest.mark.parametrize("world", [2, 3])
def test_index_exchange_gloo(lib, oracle, world):
    run_world("dist_worker.py", world)


@pytest.mark.parametrize("world", [2, 3])
def test_overlap_exchange_gloo(lib, oracle, world):
    """All-vs-all across ranks: packed read shards all-gathered, probe entries all-gathered, target shards walked per
    rank and merged (overlap_probes -> all_gather_entries -> shard_range)."""
    run_world("dist_overlap_worker.py", world)
