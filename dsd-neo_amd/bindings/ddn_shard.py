"""Work distribution for many-channel batches over the GPUs of one node.

Channels are independent streams, so the data path needs no collective: rank r owns a contiguous block of the channel
index and keeps all of those channels' carried state on its GPU.  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only for (a) broadcasting the batch descriptor from rank 0 and (b) gathering small per-channel results
or counters to rank 0.  Plumbing, not product."""
import torch
import torch.distributed as dist


def channel_range(rank, world, n_channels):
    """Block partition of [0, n_channels): the first n_channels % world ranks get one extra channel."""
    base, extra = divmod(n_channels, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def broadcast_descriptor(desc, src=0):
    """Broadcast a small picklable batch descriptor (channel count, sample count, block length, profile ...)."""
    box = [desc]
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast_object_list(box, src=src)
    return box[0]


def gather_channel_major(local, n_channels, dst=0):
    """Gather per-channel rows (tensor [local_channels, ...]) from every rank to `dst` in channel order.
    Returns the full [n_channels, ...] tensor on dst, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [channel_range(r, world, n_channels)[1] for r in range(world)]
    width = max(counts)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    dist.gather(pad, None, dst=dst)
    return None


def reduce_max_seconds(dt, device):
    """Max over ranks of a wall-clock duration (the bench contract's timing rule)."""
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
