"""The N>1 path on CPU: world_size-2 gloo.  Each rank runs the round trip on its channel shard (flan_amd.sharding), the
shards are all-gathered in rank order and must equal the unsharded result BIT FOR BIT (P5): channel sharding changes
nothing because no state crosses channels (AudioPV.cpp:41,44,108,111).  The compute stand-in on CPU is the oracle; on
the GPUs bench.py runs the same sharding/gather code with the HIP kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total_channels, n, result_dir):
    import oracle_lib as O
    from flan_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sr, W, hop, dft = 48000.0, 1024, 256, 1024
    x = O.noise(total_channels, n, seed=1234)          # every rank can regenerate the job's input (counter-based noise)
    lo, hi = sharding.shard_channels(total_channels, world, rank)
    pv = O.analyze(x[lo:hi], sr, W, hop, dft)
    out, _ = O.synthesize(pv, sr, np.float32(sr) / np.float32(hop), W)
    local = torch.from_numpy(out)
    counts = [b - a for a, b in (sharding.shard_channels(total_channels, world, r) for r in range(world))]
    if len(set(counts)) == 1:
        full = sharding.gather_output(dist, local, world)
    else:
        full = sharding.gather_output_uneven(dist, local, counts)
    t = sharding.max_over_ranks(dist, 1.0 + rank, torch.device("cpu"))
    assert t == float(world)                            # slowest rank decides
    if rank == 0:
        np.save(os.path.join(result_dir, "gathered.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total_channels", [(2, 4), (2, 3), (4, 8), (4, 6)])
def test_channel_sharding_world2_gloo(tmp_path, world, total_channels):
    """channel sharding + output all-gather over gloo with 2 and 4 ranks (even and uneven shards): the gathered buffer is the
    one-process result bit for bit"""
    import oracle_lib as O
    n = 6000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total_channels, n, str(tmp_path)), nprocs=world, join=True)
    gathered = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    x = O.noise(total_channels, n, seed=1234)
    pv = O.analyze(x, 48000.0, 1024, 256, 1024)
    ref, _ = O.synthesize(pv, 48000.0, np.float32(48000.0) / np.float32(256), 1024)
    assert gathered.shape == ref.shape
    assert np.array_equal(gathered.view(np.uint32), ref.view(np.uint32))


def test_shard_ranges():
    from flan_amd import sharding
    for total in (1, 3, 8, 64, 65):
        for world in (1, 2, 4, 8):
            ranges = [sharding.shard_channels(total, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------------------------------------------------------------
# frame-range sharding (flan_amd/sharding.py): geometry against the oracle, the overlap exchange over gloo
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("W,hop,dft,n,world", [(1024, 256, 1024, 30000, 3), (512, 128, 512, 9000, 4), (1024, 300, 1024, 20011, 2)])
def test_frame_range_analysis_slices_match_the_whole(W, hop, dft, n, world):
    """analysis_slice: a rank that analyses only ITS samples gets its frames of the whole PV, bit for bit (checked on the oracle)"""
    import oracle_lib as O
    from flan_amd import sharding as S
    x = O.noise(2, n, seed=W + hop)
    full = O.analyze(x, 48000.0, W, hop, dft)
    F = full.shape[1]
    for fb, fe in S.frame_ranges(F, world):
        s0, s1, j0 = S.analysis_slice(n, hop, W, fb, fe, F)
        assert 0 <= s0 < s1 <= n
        local = O.analyze(np.ascontiguousarray(x[:, s0:s1]), 48000.0, W, hop, dft)
        piece = local[:, j0:j0 + (fe - fb)]
        assert np.array_equal(piece.view(np.uint32), full[:, fb:fe].view(np.uint32)), (fb, fe)


def test_fold_carry_matches_sequential_fold():
    from flan_amd import sharding as S
    import math
    rng = np.random.default_rng(0)
    totals = [rng.uniform(0, S.PI2, (2, 17)) for _ in range(5)]
    carries = S.fold_carry(totals)
    run = np.zeros((2, 17))
    for t, c in zip(totals, carries):
        assert np.array_equal(c, run)
        run = np.vectorize(lambda v: math.fmod(v, S.PI2) if v > S.PI2 else v)(run + t)
    assert (np.array(carries) <= S.PI2 + 1e-12).all()


def _overlap_worker(rank, world, port, result_dir):
    from flan_amd import sharding as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hop, W, F, ch = 64, 256, 50, 2
    pad = S.pad_frames(W, hop)
    ranges = S.frame_ranges(F, world)
    rng = np.random.default_rng(77)                                   # every rank draws every rank's local output: same numbers everywhere
    locals_ = [rng.uniform(-1, 1, (ch, (fe - fb + 2 * pad) * hop)).astype(np.float32) for fb, fe in ranges]
    expect = np.zeros((ch, F * hop), np.float32)
    for (fb, fe), lo in zip(ranges, locals_):
        S.place_local_output(expect, lo, fb, hop, pad)
    fb, fe = ranges[rank]
    own = S.exchange_overlaps(dist, torch.from_numpy(locals_[rank].copy()), rank, world, hop, pad).numpy()
    # two contributions per sample at most, so the sum is the same in either order: exact
    assert np.array_equal(own, expect[:, hop * fb:hop * fe]), rank
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_range_overlap_exchange_gloo(tmp_path, world):
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)


def _short_range_worker(rank, world, port, result_dir):
    from flan_amd import sharding as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hop, W, ch = 64, 256, 1
    pad = S.pad_frames(W, hop)                                        # 2 frames: the overlap zone is 128 samples
    rows = [1, 3, 3][rank]                                            # rank 0's range is SHORTER than the zone, the others' are not
    local = torch.zeros((ch, (rows + 2 * pad) * hop))
    raised = False
    try:
        S.exchange_overlaps(dist, local, rank, world, hop, pad)
    except ValueError:
        raised = True
    with open(os.path.join(result_dir, "rank%d" % rank), "w") as f:
        f.write("1" if raised else "0")
    dist.barrier()                                                    # every rank gets here: nobody is stuck in a point-to-point wait
    dist.destroy_process_group()


def test_overlap_exchange_refuses_collectively(tmp_path):
    """a frame range shorter than the overlap zone on ONE rank: every rank raises (the decision is an all-reduce), none hangs"""
    world = 3
    mp.spawn(_short_range_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert [open(os.path.join(str(tmp_path), "rank%d" % r)).read() for r in range(world)] == ["1", "1", "1"]
