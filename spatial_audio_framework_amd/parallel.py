"""One process per GPU: how the independent units of the path (decoder / encoder instances, scenes, HRTF sets) are
sharded over ranks and how per-rank measurements are combined.

Independent instances have no exchange step: every instance owns its state, so ranks never communicate on
the data path.  `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests) is used
for the timing barriers, the MAX of the elapsed time and the gathering of results / checksums.

The one real exchange the path has is the single-scene case: when ONE sound field is fed by more sources
than an instance takes (64) and the sources are sharded over ranks, every stage is linear, so each rank renders its
sources and the ranks' loudspeaker (or SH) blocks are summed: `sum_partial_fields` = one reduce of [channels x samples]
fp32 per call (128 KiB per 512-sample block of 64 channels; xGMI is point-to-point, so blocks are summed per call of
many blocks, not per block).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for a single process)."""
    world, rank, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return world, rank, local_rank


def shard(n_units, world, rank):
    """Contiguous block of the `n_units` independent units owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (the elapsed time of the slowest rank defines the job's throughput)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def sum_partial_fields(t, root=None):
    """In-place sum over ranks of the partial sound fields `t` (a tensor on this rank's device: [.., channels, samples]).
    root = None: every rank receives the sum (all-reduce); root = r: only rank r does (reduce — half the traffic; the
    other ranks' buffers are left undefined).  Single process: no-op."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    if root is None or (t.is_cuda and dist.get_backend() == "gloo"):      # gloo has no device-tensor reduce
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
    return t


def gather_arrays(local, device="cpu"):
    """All ranks' equally shaped float32 arrays, concatenated along axis 0 in rank order (results / checksums only)."""
    import numpy as np
    if not dist.is_initialized():
        return np.asarray(local)
    t = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float32), device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()


def finalize():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
