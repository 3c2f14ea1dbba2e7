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


# ---- per-frame output record: what a rank ships to rank 0 for every frame --------------------------------------
# [ uint8 frame H*W*3 | max_inst bit-packed masks, ceil(H*W/8) bytes each (np.packbits little-endian; unused slots zero) | 8 bytes:
#   instance count (little-endian int64) ]  -- one flat uint8 tensor per frame, equal size on every rank, so ONE gather moves it.
def record_layout(H, W, max_inst):
    """(frame_bytes, mask_bytes_per_instance, record_bytes)"""
    fb, mb = H * W * 3, (H * W + 7) // 8
    return fb, mb, fb + max_inst * mb + 8


def write_record(rec, frame_u8, masks_bool, H, W, max_inst, pack_fn=None):
    """fill the flat uint8 record `rec` (device or host) from a frame [H,W,3] and boolean masks [n,H,W] (n may be 0 or exceed
    max_inst: the first max_inst are shipped, the count says how many there were).  pack_fn(mask_plane_u8, out_bytes) packs on the
    device (csm_pack_mask_bits); None = torch reference packing (CPU tests)."""
    fb, mb, _ = record_layout(H, W, max_inst)
    rec[:fb].copy_(frame_u8.reshape(-1))
    n = 0 if masks_bool is None else int(masks_bool.shape[0])
    rec[fb:fb + max_inst * mb].zero_()
    for k in range(min(n, max_inst)):
        plane = masks_bool[k].reshape(-1).to(torch.uint8)
        dst = rec[fb + k * mb: fb + (k + 1) * mb]
        if pack_fn is not None:
            pack_fn(plane, dst)
        else:
            pad = (-plane.numel()) % 8
            p8 = torch.cat([plane, plane.new_zeros(pad)]).view(-1, 8).to(torch.int32)
            dst.copy_((p8 << torch.arange(8, dtype=torch.int32, device=p8.device)).sum(1).to(torch.uint8))
    cnt = torch.tensor([n], dtype=torch.int64).view(torch.uint8)
    rec[fb + max_inst * mb:].copy_(cnt.to(rec.device))
    return rec


def read_record(rec, H, W, max_inst):
    """-> (frame uint8 [H,W,3], masks bool [min(n, max_inst), H, W], n) from a flat record (host side, rank 0)"""
    fb, mb, _ = record_layout(H, W, max_inst)
    rec = rec.cpu()
    n = int(rec[fb + max_inst * mb:].clone().view(torch.int64)[0])
    frame = rec[:fb].view(H, W, 3)
    masks = []
    for k in range(min(n, max_inst)):
        b = rec[fb + k * mb: fb + (k + 1) * mb].to(torch.int32)
        bits = ((b[:, None] >> torch.arange(8, dtype=torch.int32)) & 1).reshape(-1)[:H * W]
        masks.append(bits.bool().view(H, W))
    return frame, (torch.stack(masks) if masks else torch.zeros((0, H, W), dtype=torch.bool)), n
