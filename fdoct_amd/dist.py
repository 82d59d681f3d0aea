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


def average_bscan_over_ranks(bscan, eps, dc_mask=True):
    """The optional cross-GPU exchange of SURVEY.md 8e: every rank holds `bscan` = mean linear magnitude of ITS frames
    + eps (what fdoct_process returns, main:1220-1222), same shape (..., H, D) on every rank and the same number of
    averaged frames per rank; returns (bscan, bscandb) of the average over all ranks' frames -- one SUM all-reduce
    (RCCL over xGMI for CUDA tensors with the nccl backend) followed by the reference's log step
    20*ln(bscan)/2.303 and DC mask (main:1235-1240) on the reduced image.
    bscan: torch tensor (CPU with gloo, CUDA with nccl) or numpy array (reduced on the CPU)."""
    t = torch.as_tensor(bscan).clone()
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = (t - world * eps) / world + eps                  # mean of the per-rank means, epsilon added once
    db = 20.0 * torch.log(t) / 2.303
    if dc_mask and db.shape[-1] > 4:                      # row-major (..., H, D): depth bins 0, 1 <- bin 4
        db[..., 0] = db[..., 4]
        db[..., 1] = db[..., 4]
    if isinstance(bscan, np.ndarray):
        return t.numpy(), db.numpy()
    return t, db
