"""RCCL on the GPU box: the collectives the multi-GPU path uses (set-up broadcast of the state blob as a CUDA uint8 tensor,
MAX / SUM all-reduce of CUDA scalars, SUM all-reduce of a B-scan) on a real `nccl` (= RCCL) process group.  A gpurun box has
one GPU, so the group has one rank: this proves the RCCL plumbing of fdoct_amd/dist.py on hardware (communicator set-up,
CUDA-tensor collectives, teardown), not scaling -- the 2-rank logic is covered on CPU by tests/test_dist.py (gloo) and the
2-rank launch contract by test_bench_two_ranks_rehearsal_on_one_gpu."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_collectives_of_the_path_on_a_one_rank_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = textwrap.dedent("""
        import os, sys
        import numpy as np
        import torch
        import torch.distributed as dist
        sys.path.insert(0, %r)
        from fdoct_amd import Config, Reconstructor, synth
        from fdoct_amd import dist as fdist
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        cfg = Config(width=2048, height=8, numfftpoints=2048, numdisplaypoints=1024)
        r0 = Reconstructor(cfg)
        r0.set_background(synth.make_background(2048))
        blob = r0.export_state()
        # the same calls broadcast_state makes when world > 1, forced through RCCL on the one-rank group
        n = torch.tensor([blob.size], dtype=torch.int64, device=dev)
        dist.broadcast(n, 0)
        t = torch.from_numpy(blob).to(dev)
        dist.broadcast(t, 0)
        got = t.cpu().numpy()
        assert int(n.item()) == blob.size and (got == blob).all()
        r1 = Reconstructor(cfg)
        r1.import_state(got)
        frames = synth.make_frames(1, 2, 2048, 8)
        b0, _ = r0.process(frames)
        b1, _ = r1.process(frames)
        assert (b0 == b1).all()
        x = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        assert float(x.item()) == 1.25
        bs, db = fdist.average_bscan_over_ranks(torch.from_numpy(b0).to(dev), 1e-5)
        assert torch.allclose(bs.cpu(), torch.from_numpy(b0))
        assert fdist.max_over_ranks(0.5, dev) == 0.5 and fdist.shard_frames(10, 2, 0, 1) == (0, 10)
        dist.barrier()
        dist.destroy_process_group()
        r0.close(); r1.close()
        print("RCCL_OK")
    """ % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stderr[-2000:] + out.stdout[-500:]


def test_c_level_set_up_broadcast_over_a_caller_owned_rccl_communicator():
    """fdoct_broadcast_state_rccl (missing item 7): what a C++ host with one process per GPU calls after
    ncclCommInitRank.  Here the communicator is made through ctypes from librccl.so directly -- no torch.distributed -- with the
    one rank a gpurun box allows: the root exports, the two ncclBroadcast calls run on the handle's stream, the handle
    imports the blob it receives; a second handle given different state ends up in the root's state only through its own
    import (one rank: the call on it is a round trip of ITS state).  Errors: a null communicator, a root outside it."""
    code = textwrap.dedent("""
        import ctypes as C, sys
        import numpy as np
        sys.path.insert(0, %r)
        import torch                                   # first: its libamdhip64 is the one everything shares
        from fdoct_amd import Config, FdoctError, Reconstructor, synth
        rccl = None
        for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
            try:
                rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                pass
        assert rccl is not None, "no librccl.so on this box"
        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = UniqueId()
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
        comm = C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
        cfg = Config(width=2048, height=8, numfftpoints=2048, numdisplaypoints=1024)
        r = Reconstructor(cfg)
        yb = synth.make_background(2048)
        r.set_background(yb)
        r.set_dispersion_phase(synth.dispersion_phase(2048))
        before = r.export_state()
        frames = synth.make_frames(1, 2, 2048, 8)
        b0, _ = r.process(frames)
        r.broadcast_state_rccl(comm.value, 0)
        after = r.export_state()
        assert before.size == after.size and (before == after).all()
        b1, _ = r.process(frames)
        assert (b0 == b1).all()
        for bad in (lambda: r.broadcast_state_rccl(comm.value, 1), lambda: r.broadcast_state_rccl(0, 0)):
            try:
                bad()
                raise SystemExit("no error")
            except FdoctError:
                pass
        r.close()
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        assert rccl.ncclCommDestroy(comm) == 0
        print("RCCL_C_OK")
    """ % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0 and "RCCL_C_OK" in out.stdout, out.stderr[-2000:] + out.stdout[-500:]
