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


@pytest.mark.gpu
def test_rccl_communicator_of_the_c_abi_on_one_rank(dcn, oracle):
    """dcn_comm_* / dcn_stats_allreduce_rccl (what a host that is not Python reduces the six counters with): a
    world of ONE rank on the one GPU of the box -- RCCL is really loaded, a communicator really made, the all-reduce
    really run on its stream; with one rank the sum over ranks is the host sum over this process's contexts.  (More
    ranks need more devices: RCCL refuses two ranks on one GPU.  bench.py runs it beside torch's all-reduce at N > 1.)"""
    import numpy as np
    rng = np.random.default_rng(5)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(40, 300)))) for _ in range(500)]
    idx = dcn.Index.from_keys(np.array([1, 2, 3], np.uint64), 31, 15)
    procs = [dcn.FilterProcessor(idx, max_batch_bases=1 << 20, max_batch_reads=1 << 10) for _ in range(2)]
    b, o = oracle.concat_reads(reads)
    procs[0].filter_batch(b, o)
    procs[1].filter_batch(b, o)
    procs[1].filter_batch(b, o)
    handed = []

    def exchange(raw):
        handed.append(raw)
        return raw
    comm = dcn.distributed.Comm(1, 0, 0, exchange)
    assert len(handed[0]) == 128 and any(handed[0])
    got = comm.allreduce_counters(procs)
    one = procs[0].stats()
    assert got["total_seqs"] == 3 * len(reads) and got["total_bp"] == 3 * len(b)
    assert got == {n: 3 * one[n] for n in one}
    assert comm.allreduce_counters([]) == {n: 0 for n in one}          # a rank that had no batches
    comm.close()
    for p in procs:
        p.close()
    idx.close()


def _fake_sysfs(root, gpus):
    """gpus: [(numa, cpulist text)] -> a sysfs tree with one CPU-only KFD node followed by the GPU nodes"""
    nodes = root / "class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 96\nsimd_count 0\ndrm_render_minor 0\n")
    for i, (numa, cpus) in enumerate(gpus):
        (nodes / str(i + 1)).mkdir()
        (nodes / str(i + 1) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + i}\n")
        dev = root / "class/drm" / f"renderD{128 + i}" / "device"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{numa}\n")
        (dev / "local_cpulist").write_text(cpus + "\n")
    return str(root)


def test_ranks_are_bound_to_the_cpus_next_to_their_gpu(dcn, tmp_path, monkeypatch):
    """bench.py's N > 1 runs share one host: each rank's host threads (packers, result copies) go to the cores of its
    GPU's socket, split between the ranks of that socket (sysfs only -- no GPU call before the binding)."""
    D = dcn.distributed
    assert D.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11] and D.parse_cpulist("") == []
    assert D.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11" and D.format_cpulist([]) == ""
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    sysfs = _fake_sysfs(tmp_path, [(0, "0-47,96-143")] * 4 + [(1, "48-95,144-191")] * 4)
    topo = D.gpu_numa_topology(sysfs)
    assert [n for n, _ in topo] == [0, 0, 0, 0, 1, 1, 1, 1] and topo[5][1][:2] == [48, 49]
    allowed = set(range(192))
    shares = [D.cpus_for_rank(r, 8, allowed, topo)[0] for r in range(8)]
    assert all(len(s) == 24 for s in shares)
    assert sorted(c for s in shares for c in s) == list(range(192))           # a partition of the host
    assert set(shares[0]) <= set(topo[0][1]) and set(shares[7]) <= set(topo[7][1])
    # a cgroup that only allows part of a socket: the share comes from what is allowed
    s, note = D.cpus_for_rank(4, 8, set(range(40, 64)), topo)
    assert s == [48, 49, 50, 51] and "NUMA node 1" in note
    # nothing local is allowed / unknown GPU: no binding rather than a wrong one
    assert D.cpus_for_rank(0, 8, set(range(48, 96)), topo)[0] is None
    assert D.cpus_for_rank(9, 16, allowed, topo)[0] is None
    # one rank on a one-GPU box keeps every allowed local core
    one = D.gpu_numa_topology(_fake_sysfs(tmp_path / "one", [(0, "0-15")]))
    assert D.cpus_for_rank(0, 1, set(range(8)), one)[0] == list(range(8))
    # a rehearsal with every rank on GPU 0 (profiles/rehearse_two_ranks.sh): the ranks split GPU 0's cores
    a, b = (D.cpus_for_rank(r, 2, set(range(16)), one, gpu_of_rank=[0, 0])[0] for r in (0, 1))
    assert a == list(range(8)) and b == list(range(8, 16))
    # visible-device remapping: ordinal 0 is the second physical GPU
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,1")
    assert [n for n, _ in D.gpu_numa_topology(sysfs)] == [1, 0]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # the real call never raises, whatever this machine's sysfs holds
    before = os.sched_getaffinity(0)
    r = D.bind_rank_to_gpu_cpus(0, 1, sysfs=str(tmp_path / "absent"))
    assert r["cpus"] is None and os.sched_getaffinity(0) == before
