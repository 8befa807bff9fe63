"""bench.py --gpus N launches its own N ranks (torch.distributed.run, 127.0.0.1) before anything touches a GPU; covered here on CPU
with --plan-only (gloo rendezvous, sharding arithmetic, one JSON line from rank 0).  Also the chunked all-gather bench.py overlaps
with the compute (flan_amd.sharding.gather_chunk_into) over gloo."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("gpus", [1, 2, 4])
def test_bench_spawns_its_ranks(gpus):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--plan-only", "--channels", "8", "--seconds", "60"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # exactly one JSON line, from rank 0
    plan = json.loads(lines[0])
    assert plan["n_gpus"] == gpus and plan["scaling"] == "weak" and plan["total_channels"] == 8 * gpus
    assert plan["frames_per_step"] == gpus * 8 * (60 * 48000 // 512 + 1)
    assert [p["rank"] for p in plan["ranks"]] == list(range(gpus))
    assert [p["channels"] for p in plan["ranks"]] == [[8 * r, 8 * r + 8] for r in range(gpus)]
    if gpus > 1:
        # the output reassembly: one in-place all-gather, every rank's shard already at its final offset; and BASELINE config 4's shape (8 ch x 600 s per rank)
        out_len = (60 * 48000 // 512 + 1) * 512
        ag = plan["allgather"]
        assert ag["count_per_rank"] == 8 * out_len and ag["send_offsets"] == [r * 8 * out_len for r in range(gpus)] and ag["total_bytes"] == 4 * 8 * out_len * gpus
        c4 = plan["config4"]
        F4 = 600 * 48000 // 512 + 1
        assert c4["channels_per_gpu"] == 8 and c4["frames_per_step"] == gpus * 8 * F4
        assert c4["allgather"]["bytes_per_rank"] == 8 * F4 * 512 * 4 and c4["legs"] == ["compute_only", "compute_then_allgather", "overlapped"]
    else:
        assert plan["allgather"] is None and plan["config4"] is None


def test_a_failing_rank_fails_the_launch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    # without --plan-only the ranks need a GPU: here there is none, every rank exits non-zero and so must the launcher
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _gather_worker(rank, world, port, ch_local, n):
    from flan_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = (torch.arange(ch_local * n, dtype=torch.float32).reshape(ch_local, n) + 1000.0 * rank)
    final = torch.full((world * ch_local, n), -1.0)
    reqs = []
    k = ch_local // 2
    for c0 in (0, k):                                                  # two chunks, like bench.py
        reqs += sharding.gather_chunk_into(dist, final, local[c0:c0 + k].contiguous(), rank, world, ch_local, c0)
    for r in reqs:
        r.wait()
    want = torch.cat([torch.arange(ch_local * n, dtype=torch.float32).reshape(ch_local, n) + 1000.0 * r for r in range(world)], dim=0)
    assert torch.equal(final, want)
    assert torch.equal(final, sharding.gather_output(dist, local, world))          # the same buffer as the plain all-gather
    dist.barrier()
    dist.destroy_process_group()


def _inplace_worker(rank, world, port, ch_local, n):
    from flan_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = sharding.inplace_allgather_plan(world, ch_local, n)
    final = torch.full((world * ch_local * n,), -1.0)
    lo = plan["send_offsets"][rank]
    mine = final[lo: lo + plan["count_per_rank"]]
    mine.copy_(torch.arange(ch_local * n, dtype=torch.float32) + 1000.0 * rank)            # the synthesis writes the rank's slot of the final buffer
    # what ncclAllGather( recv + rank * count, recv, count ) does, spelled with gloo: every slot from its owner, the own one left in place
    slots = [final[o: o + plan["count_per_rank"]] for o in plan["send_offsets"]]
    dist.all_gather(slots, mine.clone())
    want = torch.cat([torch.arange(ch_local * n, dtype=torch.float32) + 1000.0 * r for r in range(world)])
    assert torch.equal(final, want)
    rows = final.view(world * ch_local, n)
    for r, (a, b) in enumerate(plan["channel_rows"]):                                     # ... and the buffer IS channel-major: rank r's channels are rows [a, b)
        assert torch.equal(rows[a:b].reshape(-1), torch.arange(ch_local * n, dtype=torch.float32) + 1000.0 * r)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_inplace_allgather_plan_is_the_final_layout(world):
    mp.spawn(_inplace_worker, args=(world, _free_port(), 4, 129), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 4])
def test_chunked_gather_is_the_all_gather(world):
    mp.spawn(_gather_worker, args=(world, _free_port(), 4, 257), nprocs=world, join=True)
