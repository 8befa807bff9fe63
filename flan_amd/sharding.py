"""Channel sharding of the phase-vocoder path across the GPUs of one node.

Channels are independent (the reference resets its per-bin state per channel: Conversions/AudioPV.cpp:41,44,108,111),
so rank r of N owns a contiguous block of channels and runs the whole path on it with no communication.  Because audio
is channel-major (float[channel][frame], AudioBuffer.cpp:479-482) the all-gather of the per-rank outputs, ordered by
rank, IS the final buffer: no re-layout.  Pure host logic, usable with any torch.distributed backend (RCCL on the GPUs,
gloo in the CPU tests).
"""


def shard_channels(total_channels, world_size, rank):
    """Contiguous channel range [lo, hi) of `rank`; the first (total % world) ranks get one extra channel."""
    base, extra = divmod(total_channels, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(dist, seconds, device):
    """Job time = the slowest rank's time."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_output(dist, local_out, world_size):
    """All-gather equal-sized per-rank outputs [ch_local][frames] into [world*ch_local][frames] (the final layout)."""
    import torch
    gathered = torch.empty((world_size * local_out.shape[0], local_out.shape[1]), dtype=local_out.dtype, device=local_out.device)
    dist.all_gather_into_tensor(gathered, local_out.contiguous())
    return gathered


def gather_output_uneven(dist, local_out, channel_counts):
    """Same for unequal shards (total channels not divisible by the world size): one broadcast per rank."""
    import torch
    parts = []
    for r, c in enumerate(channel_counts):
        buf = local_out if r == dist.get_rank() else torch.empty((c, local_out.shape[1]), dtype=local_out.dtype, device=local_out.device)
        dist.broadcast(buf, src=r)
        parts.append(buf)
    return torch.cat(parts, dim=0)
