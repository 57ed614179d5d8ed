"""Source sharding across GPUs and the sensor-image reduction.

Rays are independent; the only shared state of a render is the sensor image, a sum.  Each rank
(one process per GPU) traces a contiguous block of light-field sources into a private image; one
sum-reduce onto rank 0 finishes the job (RCCL over xGMI when the tensors live on GPUs, gloo in the
CPU tests).  The reference has no multi-GPU path (parallel_ray_tracing.cu uses device 0 only);
its chunk loop over sources (.cu:3515-3558) is what is being distributed.
"""
from __future__ import annotations


def shard_range(n_sources: int, rank: int, world_size: int):
    """Contiguous, balanced [begin, end) of rank's sources (all sources carry equal ray counts)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(int(n_sources), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_image(image, dst: int = 0):
    """Sum every rank's private image onto rank `dst` (in place).  `image` is a torch tensor
    (cuda -> RCCL ncclReduce, cpu -> gloo).  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)
    return image
