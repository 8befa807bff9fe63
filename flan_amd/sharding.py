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


def min_over_ranks(dist, value, device):
    """The smallest of the ranks' values (e.g. an "all of us succeeded" flag: 1 only if every rank says 1)."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def gather_output(dist, local_out, world_size):
    """All-gather equal-sized per-rank outputs [ch_local][frames] into [world*ch_local][frames] (the final layout)."""
    import torch
    gathered = torch.empty((world_size * local_out.shape[0], local_out.shape[1]), dtype=local_out.dtype, device=local_out.device)
    dist.all_gather_into_tensor(gathered, local_out.contiguous())
    return gathered


def inplace_allgather_plan(world_size, ch_local, samples):
    """The in-place all-gather that reassembles the output (flanhip_allgather_audio / ncclAllGather): rank r's shard of
    ch_local x samples floats sits at element offset r * count of the final float[world * ch_local][samples] buffer, count = ch_local * samples --
    the send buffer IS recv + rank * count, so the gathered buffer is the final channel-major AudioBuffer with no re-layout."""
    count = ch_local * samples
    return {"count_per_rank": count, "bytes_per_rank": 4 * count, "total_bytes": 4 * count * world_size,
            "send_offsets": [r * count for r in range(world_size)],
            "channel_rows": [[r * ch_local, (r + 1) * ch_local] for r in range(world_size)]}


def gather_chunk_into(dist, final, local_chunk, rank, world_size, ch_local, c0):
    """The all-gather of ONE channel chunk, written straight into the final layout: rank r's channels [c0, c0 + k) of its
    ch_local land in final[r * ch_local + c0 : r * ch_local + c0 + k].  One batch of point-to-point operations (every slice is
    contiguous in the channel-major buffer), so a rank can issue it for chunk i on a side stream while it still computes
    chunk i + 1 (bench.py; SURVEY 8e: the gather is per-link bound and should hide behind the compute).  Returns the requests."""
    k = local_chunk.shape[0]
    final[rank * ch_local + c0: rank * ch_local + c0 + k].copy_(local_chunk)
    ops = []
    for peer in range(world_size):
        if peer == rank:
            continue
        ops.append(dist.P2POp(dist.isend, local_chunk, peer))
        ops.append(dist.P2POp(dist.irecv, final[peer * ch_local + c0: peer * ch_local + c0 + k], peer))
    return dist.batch_isend_irecv(ops) if ops else []


def gather_output_uneven(dist, local_out, channel_counts):
    """Same for unequal shards (total channels not divisible by the world size): one broadcast per rank."""
    import torch
    parts = []
    for r, c in enumerate(channel_counts):
        buf = local_out if r == dist.get_rank() else torch.empty((c, local_out.shape[1]), dtype=local_out.dtype, device=local_out.device)
        dist.broadcast(buf, src=r)
        parts.append(buf)
    return torch.cat(parts, dim=0)


# ---------------------------------------------------------------------------------------------------------------
# Frame-range sharding (secondary axis): few channels, long signals.  Rank r owns the PV frames [fb, fe) of EVERY
# channel.  Pure geometry + exchange logic here; the device calls are flanhip_analyze_dev, flanhip_synthesize_prepass_dev and
# flanhip_synthesize_dev_carry (include/flanhip.h).
# ---------------------------------------------------------------------------------------------------------------

PI2 = 6.2831854820251465          # the reference's float constant pi2 (defines.h:44-45), as the double the device folds by


def frame_ranges(total_frames, world_size):
    """Contiguous frame ranges [fb, fe) per rank, the first (total % world) ranks one frame longer."""
    return [shard_channels(total_frames, world_size, r) for r in range(world_size)]


def pad_frames(window, hop):
    """silent frames on either side of a range so that every sample its frames reach lies inside the local output"""
    return -(-(window // 2) // hop)


def analysis_slice(n, hop, window, fb, fe, total_frames):
    """Which samples rank needs to analyse frames [fb, fe) on its own, and where they land in the local PV.

    Frame t reads samples [hop t - W/2, hop t + W/2) and the phases of frame t - 1.  The slice starts a few frames early so that
    local frame (fb - 1 - g0) has its whole window inside it: its phases are then right, and every frame after it is right in
    both m and f.  Returns (s0, s1, j0): analyse x[:, s0:s1] as a signal of its own; rows [j0, j0 + fe - fb) of the result are
    the frames [fb, fe) of the whole signal, bit for bit."""
    q = pad_frames(window, hop)
    g0 = max(fb - 1 - q, 0)
    s0 = hop * g0
    s1 = n if fe >= total_frames else min(n, hop * (fe - 1) + window)
    return s0, s1, fb - g0


def fold_carry(totals):
    """The running phase on entry to each rank from the per-rank totals (arrays [ch][bins], float64), folded like
    phase_vocoder.cpp:59.  Returns one array per rank; rank 0 gets zeros."""
    import math
    import numpy as np
    fold = np.vectorize(lambda v: math.fmod(v, PI2) if v > PI2 else v, otypes=[np.float64])
    carries, run = [], np.zeros_like(totals[0], dtype=np.float64)
    for t in totals:
        carries.append(run.copy())
        run = fold(run + np.asarray(t, np.float64))
    return carries


def place_local_output(global_out, local_out, fb, hop, pad):
    """Add a rank's local synthesis output (frames padded by `pad` silent frames on either side, so it starts at global sample
    hop (fb - pad)) into the job's output [ch][F hop]; what falls outside is dropped, like the reference drops it."""
    start = hop * (fb - pad)
    length = local_out.shape[1]
    lo, hi = max(start, 0), min(start + length, global_out.shape[1])
    if hi > lo:
        global_out[:, lo:hi] += local_out[:, lo - start:hi - start]
    return global_out


def exchange_overlaps(dist, local_out, rank, world_size, hop, pad):
    """Multi-process form of place_local_output: neighbours swap the 2 pad hop samples around their common boundary and add what
    they receive; returns this rank's own samples [pad hop, len - pad hop) completed.  local_out: tensor [ch][(rows + 2 pad) hop]."""
    import torch
    z = pad * hop
    own = local_out[:, z:local_out.shape[1] - z].clone()
    # a rank's padding zone must not reach beyond its immediate neighbour's own samples: every range holds at least `pad` frames
    # (frame_ranges() hands out contiguous ranges; with fewer frames than that per rank, shard over fewer ranks)
    # The decision is COLLECTIVE: ranges differ by one frame, so a per-rank check would raise on the short ranks only and leave their
    # neighbours waiting in batch_isend_irecv for a peer that has gone.  Every rank learns the job's shortest range first (one small
    # all-reduce) and all of them raise together, before any point-to-point operation is posted.
    shortest = torch.tensor([own.shape[1]], dtype=torch.int64, device=local_out.device)
    if world_size > 1:
        dist.all_reduce(shortest, op=dist.ReduceOp.MIN)
    if int(shortest.item()) < z:
        raise ValueError("exchange_overlaps: the job's shortest frame range (%d samples) is shorter than the overlap zone of %d: use fewer ranks" % (int(shortest.item()), z))
    ops, recv_left, recv_right = [], None, None
    if rank > 0:                                                   # my left padding zone belongs to rank - 1's last samples
        recv_left = torch.empty((local_out.shape[0], z), dtype=local_out.dtype, device=local_out.device)
        ops.append(dist.P2POp(dist.isend, local_out[:, :z].contiguous(), rank - 1))
        ops.append(dist.P2POp(dist.irecv, recv_left, rank - 1))
    if rank < world_size - 1:
        recv_right = torch.empty((local_out.shape[0], z), dtype=local_out.dtype, device=local_out.device)
        ops.append(dist.P2POp(dist.isend, local_out[:, local_out.shape[1] - z:].contiguous(), rank + 1))
        ops.append(dist.P2POp(dist.irecv, recv_right, rank + 1))
    # one batch: with RCCL, unbatched isend-then-irecv on every rank is the send-first pattern that deadlocks once a message no
    # longer fits the channel FIFO
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    if recv_left is not None:                                      # rank - 1's right padding zone = my first z samples
        own[:, :z] += recv_left
    if recv_right is not None:                                     # rank + 1's left padding zone = my last z samples
        own[:, own.shape[1] - z:] += recv_right
    return own
