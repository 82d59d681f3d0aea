"""Multi-GPU plumbing: frames are independent once the constant
state is fixed, so the path shards by contiguous frame ranges with NO data-path
collective.  The only exchange is a set-up broadcast of the constant state blob
(background, pi/dark frames, window, resample table, phase) from rank 0 -- RCCL
over xGMI when the process group is "nccl", gloo in the CPU tests -- and a MAX
reduction of the per-rank elapsed time for reporting.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_frames(nframes_total, averages, rank, world):
    """Contiguous [start, stop) frame range of `rank`; averaging groups never straddle ranks."""
    groups = nframes_total // averages
    base, extra = divmod(groups, world)
    g0 = rank * base + min(rank, extra)
    g1 = g0 + base + (1 if rank < extra else 0)
    return g0 * averages, g1 * averages


def broadcast_state(blob, src=0, device=None):
    """blob: uint8 numpy array on `src` (ignored elsewhere).  Returns the blob on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.ascontiguousarray(blob, np.uint8)
    dev = device if device is not None else torch.device("cpu")
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    if dist.get_rank() == src:
        n[0] = int(blob.size)
    dist.broadcast(n, src)
    t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(blob, np.uint8)))
    dist.broadcast(t, src)
    return t.cpu().numpy()


def max_over_ranks(seconds, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
