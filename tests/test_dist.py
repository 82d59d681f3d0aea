"""Multi-GPU plumbing on CPU (gloo, world_size 2): contiguous frame shards that never split an
averaging group, the set-up broadcast of the constant-state blob (the only collective of the
path) and the MAX-over-ranks timing reduction bench.py reports."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from fdoct_amd import dist as fdist


def test_shard_frames_partition():
    for total, A, world in [(80000, 1, 8), (160, 16, 8), (100, 4, 3), (7, 1, 2), (16, 16, 4)]:
        got = [fdist.shard_frames(total, A, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == (total // A) * A
        for (a0, a1), (b0, b1) in zip(got, got[1:]):
            assert a1 == b0
        for s, e in got:
            assert s % A == 0 and e % A == 0 and e >= s
        sizes = [e - s for s, e in got]
        assert max(sizes) - min(sizes) <= A


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = np.arange(1000, dtype=np.uint8) * 3 if rank == 0 else np.zeros(0, np.uint8)
    got = fdist.broadcast_state(blob, 0)
    t = fdist.max_over_ranks(0.5 + rank)
    s = fdist.sum_over_ranks(10.0 * (rank + 1))
    s0, s1 = fdist.shard_frames(101, 1, rank, world)
    q.put((rank, got.tobytes(), t, s, s1 - s0))
    dist.barrier()
    dist.destroy_process_group()


def test_state_broadcast_and_reductions_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = (np.arange(1000, dtype=np.uint8) * 3).tobytes()
    assert res[0][1] == want and res[1][1] == want
    assert res[0][2] == res[1][2] == 1.5
    assert res[0][3] == res[1][3] == 30.0
    assert res[0][4] + res[1][4] == 101


def _avg_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    mags = rng.random((world, 3, 5, 16)) + 0.1           # per-rank mean magnitudes (2 outputs... any leading shape)
    eps = 1e-5
    b, d = fdist.average_bscan_over_ranks(mags[rank] + eps, eps)
    q.put((rank, b.tobytes(), d.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_cross_rank_bscan_average_gloo_world2():
    """SURVEY 8e's optional reduce: the average over ranks of per-rank averaged B-scans, then the reference's log step."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_avg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(7)
    mags = rng.random((2, 3, 5, 16)) + 0.1
    want_b = mags.mean(axis=0) + 1e-5
    want_d = 20.0 * np.log(want_b) / 2.303
    want_d[..., 0] = want_d[..., 4]
    want_d[..., 1] = want_d[..., 4]
    for r in res:
        np.testing.assert_allclose(np.frombuffer(r[1]).reshape(want_b.shape), want_b, rtol=1e-14)
        np.testing.assert_allclose(np.frombuffer(r[2]).reshape(want_d.shape), want_d, rtol=1e-13, atol=1e-13)


def test_c_abi_shard_rule_equals_the_python_one():
    """fdoct_shard_frames (what the C++ host's --gpus mode uses) is the same partition as dist.shard_frames: pure host
    arithmetic, callable without a GPU."""
    import pytest
    from fdoct_amd import FdoctError, capi
    for total, A, world in [(80000, 1, 8), (160, 16, 8), (100, 4, 3), (7, 1, 2), (16, 16, 4), (0, 1, 3), (33, 2, 5)]:
        for r in range(world):
            assert capi.shard_frames(total, A, r, world) == fdist.shard_frames(total, A, r, world)
    with pytest.raises(FdoctError):
        capi.shard_frames(10, 1, 3, 3)
    with pytest.raises(FdoctError):
        capi.shard_frames(10, 0, 0, 1)
    assert capi.load_library().fdoct_device_count() >= 0


def test_bench_launches_its_own_ranks_as_child_processes(monkeypatch):
    """`python3 bench.py --gpus N` without a launcher (the way the driver calls `--gpus 1`): bench.py must start
    torch.distributed.run as a CHILD process -- never replace itself -- with one rank per GPU on 127.0.0.1, pass its own
    arguments through, and hand the child's exit code back.  (The launch itself runs on the GPU box:
    tests/test_gpu_c5.py::test_bench_direct_launch_*.)"""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module("bench")
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"], seen["kw"] = cmd, env, kw
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert os.path.basename(cmd[-7]) == "bench.py" and cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # main() takes that path before it imports torch or touches a GPU
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n: calls.append(n) or 0)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [4]


def test_bench_process_group_check(monkeypatch):
    """bench.py refuses (exit code 3, no JSON line) a process group that is not N ranks on N devices: the rule itself, on the
    CPU (its use under a launcher runs on the GPU box: tests/test_gpu_c5.py::test_bench_refuses_a_world_size_that_is_not_gpus)."""
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module("bench")
    ok = bench.process_group_problems
    assert ok(1, 1, ["0000:05:00.0"], False) == []
    assert ok(8, 8, ["0000:%02x:00.0" % (5 + i) for i in range(8)], False) == []
    assert ok(2, 2, ["0000:05:00.0", "0000:05:00.0"], True) == []                      # the rehearsal switch
    assert any("one device" in m for m in ok(2, 2, ["0000:05:00.0", "0000:05:00.0"], False))
    assert any("holds 2 ranks" in m for m in ok(2, 4, ["a", "b"], False))
    assert ok(2, 2, [None, None], False) == []                                          # PCI addresses unknown: nothing to compare
    assert any("device entries" in m for m in ok(2, 2, ["a"], False))
