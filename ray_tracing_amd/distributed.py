"""Multi-GPU rendering: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Every (pixel, sample) path of the reference's loop (src/main.rs:570-584) is independent and the only
coupling is the per-pixel sum (src/main.rs:585-589), so the work shards by SAMPLE RANGE: rank r renders
samples [offset_r, offset_r + count_r) of every pixel into its own accumulation buffer, and one
reduce(SUM) of W*H*3 floats to rank 0 finishes the image.  The Philox counter carries the absolute
sample index, so the union of the shards is the same set of paths a single GPU would trace.
"""
import torch
import torch.distributed as dist


def shard_samples(total_spp, rank, world_size):
    """Contiguous, balanced split of range(total_spp): returns (first_sample, n_samples) of `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(total_spp), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def reduce_accumulators(accum, dst=0, group=None):
    """Sums the ranks' accumulation buffers into rank `dst` (in place) -- the in-process channel reduce of
    src/main.rs:585-589 lifted across GPUs.  `accum` is a torch tensor on this rank's device."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            # gloo has no reduce for device tensors (a debugging set-up: several ranks on one GPU, or no RCCL): through the host
            host = accum.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            accum.copy_(host)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


def render_sharded(render_fn, height, width, total_spp, device, sample_base=0, group=None):
    """render_fn(accum_tensor, first_sample, n_samples) must ADD the colour sums of that sample range into
    accum_tensor ([H, W, 3] float32 on `device`).  Returns the rank-local tensor; after the call rank 0's
    tensor holds the whole image's sums."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    first, count = shard_samples(total_spp, rank, world)
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device=device)
    if count > 0:
        render_fn(accum, sample_base + first, count)
    return reduce_accumulators(accum, 0, group)


def render_tile_sharded(render_fn, height, width, device, group=None):
    """Pixel sharding, for renders of fewer samples than GPUs (the reference's outer loops run over pixels, src/main.rs:567-568).
    render_fn(accum_tensor, shard, n_shards) must ADD every sample of pixel shard `shard` of `n_shards` -- the row bands that
    ray_tracing_amd.shard_tiles / rt_shard_tiles select -- into accum_tensor ([H, W, 3] float32 on `device`) and leave the other
    pixels alone.  The shards are disjoint, so the reduce(SUM) to rank 0 only adds zeros to each pixel: rank 0's image equals the
    one-GPU render bit for bit."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device=device)
    render_fn(accum, rank, world)
    return reduce_accumulators(accum, 0, group)
