"""N>1 path on CPU: world_size 2, backend gloo.  Shard assignment, counter all-reduce and the ordered merge of keep
bitmaps must reproduce the single-process result exactly."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_batches_of_rank_partition(dcn):
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(b for r in range(world) for b in dcn.distributed.batches_of_rank(n, r, world))
            assert seen == list(range(n))
    assert dcn.distributed.split_units(10, 4) == [(0, 4), (4, 8), (8, 10)]


@pytest.mark.parametrize("world", [2])
def test_two_ranks_gloo(tmp_path, world):
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    res = json.load(open(out))
    assert res["world"] == world and res["n_batches"] == 16
    assert res["stats"] == res["want"]
    assert res["keep_equal"]


@pytest.mark.gpu
def test_two_ranks_with_the_gpu_engine(tmp_path):
    """the same job with the real engine on every rank (two processes sharing GPU 0, own replica + context each,
    counters from dcn_ctx_stats): decisions in input order and the all-reduced counters equal the single-process
    oracle's.  gloo carries the collectives (RCCL does not take two ranks on one device)."""
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), "gpu"]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    res = json.load(open(out))
    assert res["world"] == 2 and res["stats"] == res["want"] and res["keep_equal"]
