"""Multi-GPU layer (new functionality; the reference is single process).

The path shards by frame: independent units, no exchange during compute.  One process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests):
  * rank r of R owns frames {i : i mod R == r},
  * packed weights are broadcast once from rank 0,
  * per-frame outputs (uint8 frames / packed masks) are gathered to rank 0,
  * barriers only around timing.  No all-reduce anywhere on the data path.
"""
import torch


def frames_of_rank(n_frames, rank, world):
    """indices of the frames rank `rank` processes (animeinsseg/__init__.py:485-499 is a per-image loop)"""
    return list(range(rank, n_frames, world))


def broadcast_weights(flat, dist=None, src=0):
    """one RCCL broadcast of the packed fp32 weight buffer (about 0.9 GB for all nets) -- start-up only"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return flat
    dist.broadcast(flat, src=src)
    return flat


def gather_outputs(local, n_frames, dist=None, dst=0):
    """local: list of same-shaped tensors for frames_of_rank(...) in order -> on `dst`, the list of all
    n_frames outputs in frame order; other ranks get None.  Ranks with fewer frames pad the last round."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    world, rank = dist.get_world_size(), dist.get_rank()
    rounds = (n_frames + world - 1) // world
    proto = local[0] if local else None
    out = [None] * n_frames
    for k in range(rounds):
        if proto is None:
            raise RuntimeError("gather_outputs: a rank without frames has no prototype; pass >= world frames")
        mine = local[k] if k < len(local) else torch.zeros_like(proto)
        bucket = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, bucket, dst=dst)
        if rank == dst:
            for r in range(world):
                i = k * world + r
                if i < n_frames:
                    out[i] = bucket[r]
    return out if rank == dst else None
