"""BASELINE configs[4] ("C5": 8 x 10 000 synthetic 2048-pt x 1000-line frames sharded over 8 GPUs) on what a one-GPU box has:

  * ONE shard at its stated size -- 10 000 device-resident frames (41 GB in, 41 GB out per output) through a single
    fdoct_process_async call;
  * the launch contract `python3 bench.py --gpus N` invoked DIRECTLY (no torch.distributed.run around it): bench.py starts
    its ranks itself as child processes;
  * the entry points leave the calling thread's current HIP device as they found it (a host that drives several GPUs).

The 8-GPU run itself is the driver's (SCALE_rNN.json); frames shard with no data-path collective, so a shard
on one GPU is the whole per-GPU data path.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from fdoct_amd import DTYPE_U16, Config, Reconstructor, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c5_one_shard_of_10000_frames_through_one_call():
    import torch
    W, H, N, D = 2048, 1000, 2048, 1024
    nf, distinct = 10000, 16
    free, _ = torch.cuda.mem_get_info()
    need = nf * H * (W * 2 + 2 * D * 4) + (4 << 30)
    if free < need:
        pytest.skip("needs %.0f GB of free HBM, %.0f GB free" % (need / 1e9, free / 1e9))
    base = synth.make_frames(100, distinct + 1, W, H)                  # frame 100 + i, i <= 16
    yb = synth.make_background(W)
    dev = torch.device("cuda", 0)
    d_base = torch.from_numpy(base.view(np.int16)).to(dev)
    d_in = d_base[:distinct].repeat(nf // distinct, 1, 1)              # (10000, 1000, 2048): a small ring tiled on the device
    d_in[nf - 1] = d_base[distinct]                                    # the last frame of the shard is one of a kind
    assert d_in.shape[0] == nf and d_in.is_contiguous()
    d_b = torch.full((nf, H, D), float("nan"), dtype=torch.float32, device=dev)
    d_db = torch.full((nf, H, D), float("nan"), dtype=torch.float32, device=dev)
    r = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D))
    r.set_background(yb)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    r.set_stream(st.cuda_stream)
    r.set_timing(True)
    r.process_device(d_in.data_ptr(), DTYPE_U16, nf, W * 2, d_b.data_ptr(), d_db.data_ptr())   # ONE call, 10^7 A-scans
    r.synchronize()
    t = r.timing()
    assert t["ascans"] == nf * H and t["bytes_in"] == nf * H * W * 2 and t["bytes_out"] == 2 * nf * H * D * 4
    r.close()
    # every one of the 2 x 10^10 output floats was written and is finite
    for out in (d_b, d_db):
        for f0 in range(0, nf, 1000):
            assert bool(torch.isfinite(out[f0:f0 + 1000]).all()), "non-finite output in frames %d.." % f0
    # oracle parity on the first and last 8 A-scans of the first and the last frame (rows are independent: the oracle run
    # on those rows alone IS the reference result for them)
    cfg8 = Config(width=W, height=8, numfftpoints=N, numdisplaypoints=D)
    for f, src in ((0, base[0]), (nf - 1, base[distinct])):
        for sl in (slice(0, 8), slice(H - 8, H)):
            mag_o, _, db_o = helpers.oracle_reference(cfg8, np.ascontiguousarray(src[None, sl]), yb)
            helpers.check_mag(d_b[f, sl].cpu().numpy()[None], mag_o, "C5 shard frame %d rows %s" % (f, sl))
            helpers.check_db(d_db[f, sl].cpu().numpy()[None], np.transpose(db_o, (0, 2, 1)), mag_o, "C5 shard frame %d rows %s" % (f, sl))
    # the tiled ring: frame f repeats frame f mod 16 bit for bit, wherever in the 41 GB it lies (64-bit row offsets)
    for f in (16, 17, 4999, 5000, 8191, 8192, nf - 2):
        assert torch.equal(d_b[f], d_b[f % distinct]) and torch.equal(d_db[f], d_db[f % distinct]), f
    assert not torch.equal(d_b[nf - 1], d_b[(nf - 1) % distinct])
    # analytic peak bin (wangOCTrec4.m:200-202) on every row of a strided sample of frames
    for f in list(range(0, nf, 997)) + [nf - 1]:
        ls1, _ = synth.frame_depths_um(100 + (distinct if f == nf - 1 else f % distinct), H)
        want = synth.expected_peak_bin(ls1, W)
        got = (d_b[f][:, 3:].argmax(dim=1) + 3).cpu().numpy()
        assert np.abs(got - want).max() <= 2.5, (f, np.abs(got - want).max())


def test_bench_direct_launch_of_two_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2 ...` the way the driver calls `--gpus 1`: no torch.distributed.run around it.  bench.py
    starts one rank per GPU as child processes itself and relays rank 0's single JSON line and the exit code.  Two ranks
    share this box's one GPU over gloo (RCCL refuses two ranks on one device); on a multi-GPU node the default backend is
    nccl = RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", "3",
           "--warmup", "1", "--ramp-seconds", "0", "--frames-per-step", "6", "--no-cpu-baseline", "--stage-steps", "0",
           "--half-chip-steps", "0", "--sustained-seconds", "0.2"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["config"]["parallelism"] == "frame-shard x2"
    assert d["process_group"]["ranks_seen"] == 2 and d["process_group"]["backend"] == "gloo"
    assert len(d["process_group"]["devices"]) == 2
    assert "failed" not in d["parity"]
    assert d["cpu_baseline"] is None and "cpu_baseline_note" in d
    assert d["sustained"]["steps"] > 0
    # the line cannot be misread: `value` from the slowest rank's DEVICE time, the wall clock next to it, every
    # rank's own figures, the set-up broadcast's duration
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and all(p["device_ms_per_step"] > 0 for p in pr)
    slowest = max(p["device_ms_per_step"] for p in pr)
    assert abs(d["ms_per_step"] - slowest) <= 2e-4 + 1e-3 * slowest
    per_gpu = d["config"]["frames_per_step_per_gpu"] * d["config"]["lines_per_frame"]
    # (the per-rank figure is printed to 1e-4 ms: on a 0.02 ms step that rounding alone is 0.2 % -- it failed once at 0.21 %)
    assert abs(d["value"] - 2 * per_gpu / (slowest * 1e-3)) <= (2e-3 + 0.6e-4 / slowest) * d["value"]
    assert d["value"] <= sum(p["ascans_per_s"] for p in pr) * (1 + 1e-6)
    assert d["wall_ms_per_step"] >= d["ms_per_step"] * 0.999          # host clock around synchronize() + barrier()
    sb = d["process_group"]["setup_broadcast"]
    assert sb["bytes"] > 4 * 2048 and sb["ms"] > 0


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`--gpus 3` under a launcher that started two ranks: exit code 3 and no JSON line -- never a line that reads as a 3-GPU
    result."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29653",
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--share-gpu", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--ramp-seconds", "0",
           "--frames-per-step", "2", "--no-cpu-baseline", "--stage-steps", "0", "--half-chip-steps", "0", "--sustained-seconds", "0"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "refusing to run" in out.stderr


def test_bench_under_a_launcher_without_gpus_adopts_the_world_size():
    """`torchrun --nproc-per-node 2 bench.py` with no --gpus at all: the launcher's world size is the GPU count,
    and the line says n_gpus 2 -- only an EXPLICIT --gpus that disagrees with the launcher is refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29654",
           os.path.join(ROOT, "bench.py"), "--share-gpu", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--ramp-seconds", "0",
           "--frames-per-step", "2", "--no-cpu-baseline", "--stage-steps", "0", "--half-chip-steps", "0", "--sustained-seconds", "0", "--precise-steps", "0"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["process_group"]["ranks_seen"] == 2


def test_bench_direct_launch_failure_is_relayed():
    """The child launcher's exit code comes back: an impossible rank count for the nccl backend on this box (two ranks,
    one device, no --share-gpu) must not read as success."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the launch would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--ramp-seconds", "0",
           "--frames-per-step", "2", "--no-cpu-baseline", "--stage-steps", "0", "--half-chip-steps", "0", "--sustained-seconds", "0"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_entry_points_leave_the_callers_current_device_alone():
    """Every entry point runs on the handle's device and restores the thread's current HIP device (include/fdoct.h).  With
    two or more GPUs the handle lives on the LAST device while the thread's current device stays 0; on a one-GPU box the
    same calls are made and the current device must read 0 after each."""
    hip = ctypes.CDLL("libamdhip64.so")
    cur = ctypes.c_int(-1)

    def current():
        assert hip.hipGetDevice(ctypes.byref(cur)) == 0
        return cur.value

    from fdoct_amd import capi
    ndev = capi.load_library().fdoct_device_count()
    assert ndev >= 1
    assert hip.hipSetDevice(0) == 0
    W, H, N, D = 2048, 4, 2048, 1024
    frames, yb = synth.make_frames(0, 2, W, H), synth.make_background(W)
    r0 = Reconstructor(Config(width=W, height=H, numfftpoints=N, numdisplaypoints=D, device=0))
    r0.set_background(yb)
    b0, d0 = r0.process(frames)
    other = ndev - 1
    r1 = r0.clone_to_device(other)
    assert current() == 0
    b1, d1 = r1.process(frames)
    assert current() == 0
    np.testing.assert_array_equal(b0, b1)
    np.testing.assert_array_equal(d0, d1)
    r1.set_staged(True)
    r1.process(frames)
    r1.get_ylin(0, 2)
    assert current() == 0
    r1.set_staged(False)
    g = r1.display(np.transpose(d1, (0, 2, 1)))
    assert g.dtype == np.uint8 and current() == 0
    r1.lockin_db(b1, b1[0])
    r1.frontend(np.zeros((1, 8, 16), np.uint8), 3, 2, 2)
    r1.synchronize()
    r1.timing()
    assert current() == 0
    r1.close()
    assert current() == 0
    r0.close()


def test_handles_give_back_every_byte_of_device_and_pinned_memory():
    """An instrument program creates and drops handles as the operator changes the ini (a new fdoct_create per geometry):
    40 create / configure / process / destroy cycles over four geometries -- every kernel family, host-pointer and
    device-pointer calls, both layouts -- must leave the device's free memory where it was."""
    import torch
    from fdoct_amd import LAYOUT_TRANSPOSED
    torch.cuda.synchronize()
    shapes = [dict(width=2048, height=64, numfftpoints=2048, numdisplaypoints=1024),                                  # fused (+ its own transposed store)
              dict(width=160, height=24, numfftpoints=2560, numdisplaypoints=320, increasefftpointsmultiplier=4),     # wave per row
              dict(width=300, height=16, numfftpoints=1000, numdisplaypoints=400),                                    # any-configuration kernel
              dict(width=4096, height=4, numfftpoints=16384, numdisplaypoints=512, increasefftpointsmultiplier=4)]    # long rows (global-memory passes)

    def cycle(i):
        kw = shapes[i % len(shapes)]
        cfg = Config(**kw)
        r = Reconstructor(cfg)
        r.set_background(synth.make_background(cfg.width))
        fr = synth.make_frames(i, 2, cfg.width, cfg.height)
        b, db = r.process(fr, layout=LAYOUT_TRANSPOSED if i % 2 else 0)
        assert np.isfinite(b).all() and np.isfinite(db).all()
        r.close()

    for i in range(len(shapes)):      # first use of each family: module load, torch allocator warm-up
        cycle(i)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(40):
        cycle(i)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), "device memory shrank by %.1f MB over 40 handle lifetimes" % ((free0 - free1) / 1e6)
