#!/usr/bin/env python3
"""bench.py -- PV analysis+resynthesis frames/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic audio that is already resident in HBM:
    Audio::convert_to_PV(2048, 512, dft 2048)  ->  PV::convert_to_audio        (flanhip_analyze_dev_fused + flanhip_synthesize_dev_fused)
Workload at N=1: 8 channels x 60 s x 48 kHz uniform noise (the configuration BASELINE.json's north_star quotes its targets on:
"60 s x 8-ch 48 kHz convertToPV->convertToAudio round-trip at 1 GPU").  With --gpus N every rank owns its own 8 channels
(channels are independent: AudioPV.cpp:41,44,108,111), so the job is 8N channels, "scaling": "weak", with no collective inside
the compute.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment LAUNCHES the N ranks itself (torch.distributed.run,
one process per GPU, 127.0.0.1 rendezvous) before anything here touches a GPU; under torchrun / the driver's own launch it is
one of those ranks.  The RCCL all-gather that reassembles the output buffer (float[8N][frames], channel-major, so the gathered
buffer IS the final layout) is INSIDE the timed step at N > 1, overlapped with the next batch's compute on a side stream (the north star's
path ends with it): `value` includes it, `value_compute_only` and the not-overlapped figure stand beside it ("allgather").

--seconds 600 gives the per-GPU shard of BASELINE config 4 (64 ch x 10 min over 8 GPUs = 8 ch x 600 s each).

PyTorch here is plumbing only: device buffers, streams, events and torch.distributed.  The kernels are the HIP library
flan_amd/libflanhip.so called through the C ABI (include/flanhip.h).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 48000.0
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)


def baseline_metric():
    """the metric string exactly as BASELINE.json names it"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except Exception:
        return "PV analysis+resynthesis frames/sec (2048-win, hop 512, 48 kHz) at 1/2/4/8 GPU"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=8, help="channels per GPU")
    ap.add_argument("--seconds", type=float, default=60.0, help="60: the metric's workload; 600: the per-GPU shard of BASELINE config 4")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather-chunks", type=int, default=1, help="channel chunks per batch in the overlapped all-gather measurement (1: whole batches, double buffered)")
    ap.add_argument("--no-configs", action="store_true", help="skip the 'configs' object (BASELINE configs 3 and 5, the API-default dft 4096 call)")
    ap.add_argument("--kernel-variant", default="", help="diagnostic: WHICH=VALUE[,..] for flanhip_debug_option (A/B of kernel generations, chain lengths; never for the metric)")
    ap.add_argument("--unfused", action="store_true", help="run synthesis' pre-pass as its own kernel instead of inside analysis")
    ap.add_argument("--preroll-ms", type=float, default=80.0,
                    help="untimed device warm-up before the W warm-up steps: the same steps run for this long so that the GPU's clocks "
                         "have settled (a just-woken MI355X runs the same launch ~20 %% slower for its first ~40 ms); 0 disables")
    ap.add_argument("--pcie", action="store_true", help="(default since round 6) also time the host-buffer entry points (PCIe inclusive; reported in 'pcie_inclusive', never in 'value')")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg (SURVEY 8d: reported separately, by default)")
    ap.add_argument("--window", type=int, default=2048, help="2048: the BASELINE metric")
    ap.add_argument("--hop", type=int, default=512, help="512: the BASELINE metric; 128 with --dft 4096 is the reference API's default call")
    ap.add_argument("--dft", type=int, default=2048,
                    help="2048: the primary measurement; 4096: the literal convert_to_PV(2048,512) default of the reference API (SURVEY 8)")
    ap.add_argument("--config4-seconds", type=float, default=600.0,
                    help="N > 1: seconds per channel of the BASELINE config 4 leg (8 ch x 600 s per rank); tests rehearse it with a few seconds")
    ap.add_argument("--no-config4", action="store_true", help="N > 1: skip the config 4 leg")
    ap.add_argument("--plan-only", action="store_true",
                    help="no GPU work: the ranks rendezvous over gloo, agree on the plan and rank 0 prints it (what tests/ use to cover the launcher on CPU)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """--gpus N > 1 outside a distributed launch: start the N rank processes (one per GPU) and hand their verdict on.  Runs before
    this process has made any GPU call (a process that has initialised the GPU must not start or become another program)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def cpu_baseline(channels, seconds, threads, window, hop, dft):
    """The CPU oracle (oracle/flan_oracle.cpp, a port of the reference path) on the host cores; checker code used as the
    reported baseline only.  Returns (frames/s, frames, seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    n = int(seconds * SR)
    x = O.noise(channels, n, seed=1234)

    def one(c):
        pv = O.analyze(x[c:c + 1], SR, window, hop, dft)
        O.synthesize(pv, SR, SR / hop, window)
        return pv.shape[1]

    t0 = time.perf_counter()
    if threads <= 1:
        frames = sum(one(c) for c in range(channels))
    else:
        with ThreadPoolExecutor(threads) as ex:       # ctypes releases the GIL; one channel per task
            frames = sum(ex.map(one, range(channels)))
    dt = time.perf_counter() - t0
    return frames / dt, frames, dt


def cpu_fft_share(frames, dft, per_frame_seconds):
    """How much of the oracle's time is its own FFT (fp64 radix-2, not FFTW): r2c + c2r per frame with one plan, timed alone."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    import ctypes
    lib = O._load()
    lib.oracle_fft_pairs.restype = ctypes.c_double
    lib.oracle_fft_pairs.argtypes = [ctypes.c_int, ctypes.c_int]
    reps = max(2000, min(20000, frames // 4))
    lib.oracle_fft_pairs(dft, 200)                    # plan construction and first touch outside the clock
    t0 = time.perf_counter()
    lib.oracle_fft_pairs(dft, reps)                   # one plan, reps x (r2c + c2r): what a frame of the round trip pays
    per = (time.perf_counter() - t0) / reps
    return round(per / per_frame_seconds, 3), round(per * 1e6, 2)


class GatherLegs:
    """N > 1: the output reassembly of the north star -- ONE in-place all-gather of float[world x channels][samples] -- behind the round trip, through
    the library's C ABI (flanhip_comm_unique_id / flanhip_comm_init / flanhip_allgather_audio: ncclAllGather over xGMI, collective.hip).  The
    communicator id travels from rank 0 over the job's own torch.distributed group; if any rank cannot bind RCCL through the C ABI, every rank uses
    torch.distributed.all_gather_into_tensor instead (`api` / `err` say which)."""

    def __init__(self, fa, torch, dist, sharding, rank, world, dev, ctl_dev):
        self.fa, self.torch, self.dist, self.sharding = fa, torch, dist, sharding
        self.rank, self.world, self.dev, self.ctl_dev = rank, world, dev, ctl_dev
        self.comm, self.err = None, None
        lib = fa.lib
        uid = ctypes.create_string_buffer(128)
        ok = 1
        if rank == 0:
            rc = lib.flanhip_comm_unique_id(uid)
            if rc != 0:
                ok, self.err = 0, "flanhip_comm_unique_id: %s" % fa.last_error()
        t = torch.tensor(list(uid.raw) + [ok], dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        if raw[128]:
            comm = ctypes.c_void_p()
            rc = lib.flanhip_comm_init(ctypes.create_string_buffer(raw[:128], 128), world, rank, ctypes.byref(comm))
            if rc == 0 and comm.value:
                self.comm = comm
            else:
                ok, self.err = 0, "flanhip_comm_init: %s" % fa.last_error()
        else:
            ok = 0
        if sharding.min_over_ranks(dist, 1 if self.comm else 0, ctl_dev) < 1:
            if self.comm:
                lib.flanhip_comm_destroy(self.comm)
            self.comm = None
            self.err = self.err or "another rank could not bind RCCL through the C ABI"
        self.api = "flanhip_allgather_audio: ncclAllGather in place through the C ABI" if self.comm else "torch.distributed.all_gather_into_tensor in place (C-ABI communicator not available)"
        self.nranks = world if self.comm else dist.get_world_size()
        self.side = torch.cuda.Stream(device=dev)

    def close(self):
        if self.comm:
            self.fa.lib.flanhip_comm_destroy(self.comm)
            self.comm = None

    def legs(self, audio, pv, ws, ch, n, W, HOP, DFT, steps, warmup, timed_region, unfused=False):
        fa, torch, dist, rank, world = self.fa, self.torch, self.dist, self.rank, self.world
        lib = fa.lib
        F = int(lib.flanhip_num_pv_frames(n, HOP))
        bins, ar = DFT // 2 + 1, SR / HOP
        count = ch * F * HOP                                      # floats per rank
        finals = [torch.empty((world * ch, F * HOP), dtype=torch.float32, device=self.dev) for _ in range(2)]
        mine = [f[rank * ch: rank * ch + ch] for f in finals]    # this rank's slot: the synthesis writes it in place
        nan_flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        main = torch.cuda.current_stream()
        side = self.side
        stream = main.cuda_stream

        def compute(b):
            if unfused:
                fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv, stream)
            else:
                fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
            fa.synthesize_dev_stages(pv, ch, F, bins, SR, ar, W, mine[b], ws, nan_flag, 0 if unfused else 1, 0xF, stream)

        pending = [None, None]

        def gather_on(b, s):
            """the in-place all-gather of finals[b], behind everything issued on stream s so far"""
            if self.comm:
                fa.check(lib.flanhip_allgather_audio(self.comm, ctypes.c_void_p(finals[b].data_ptr()), count, rank, ctypes.c_void_p(s.cuda_stream)))
                ev = torch.cuda.Event()
                ev.record(s)
                return ev
            with torch.cuda.stream(s):
                return dist.all_gather_into_tensor(finals[b], mine[b], async_op=True)

        def wait_for(h):
            if h is None:
                return
            if self.comm:
                torch.cuda.current_stream().wait_event(h)
            else:
                h.wait()

        def step_then_gather():
            compute(0)
            wait_for(gather_on(0, main))

        counter = [0]

        def step_overlapped():
            b = counter[0] & 1
            counter[0] += 1
            wait_for(pending[b])                                  # the gather that last used this buffer
            compute(b)
            done = torch.cuda.Event()
            done.record(main)
            side.wait_event(done)
            pending[b] = gather_on(b, side)

        def drain():
            for b in (0, 1):
                wait_for(pending[b])
                pending[b] = None
            main.wait_stream(side)

        res = {}
        res["overlapped_s"] = timed_region(step_overlapped, drain, steps, warmup)
        # what arrived: every slot of both buffers against the sum its owner reports for its own slot
        ok = True
        for b in (0, 1):
            sums = finals[b].view(world, -1).sum(dim=1, dtype=torch.float64)
            own = sums[rank].reshape(1).clone()
            alls = [torch.empty_like(own) for _ in range(world)]
            dist.all_gather(alls, own)
            want = torch.cat(alls)
            ok = ok and bool(torch.all(torch.abs(sums - want) <= 1e-9 * torch.abs(want) + 1e-12).item()) and bool(torch.all(torch.isfinite(sums)).item())
        assert ok, "an all-gathered slot does not hold what its owner computed"
        res["slots_verified"] = ok
        res["compute_only_s"] = timed_region(lambda: compute(0), lambda: None, steps, warmup)
        res["then_gather_s"] = timed_region(step_then_gather, lambda: None, steps, warmup)
        reps = 5
        res["gather_alone_ms"] = 1e3 * timed_region(lambda: wait_for(gather_on(0, main)), lambda: None, reps, 2) / reps
        del finals, mine
        return res

    def config4(self, seconds, timed_region):
        """BASELINE config 4 (64 ch x 10 min over 8 GPUs): this rank's 8 ch x `seconds` s through the same three legs, a few steps each"""
        fa, torch, rank, world = self.fa, self.torch, self.rank, self.world
        lib = fa.lib
        W, HOP, DFT, ch = 2048, 512, 2048, 8
        n = int(seconds * SR)
        F = int(lib.flanhip_num_pv_frames(n, HOP))
        bins, ar = DFT // 2 + 1, SR / HOP
        audio = torch.empty((ch, n), dtype=torch.float32, device=self.dev)
        fa.check(lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 4321 + rank, None))
        pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=self.dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, ar, W), dtype=torch.uint8, device=self.dev)
        steps, warmup = 3, 1
        legs = self.legs(audio, pv, ws, ch, n, W, HOP, DFT, steps, warmup, timed_region)
        frames = world * ch * F
        out = {"workload": "BASELINE config 4: %d ch x %.0f s over %d GPU(s), 8 ch per rank: convert_to_PV(2048,512,2048) -> convert_to_audio -> in-place all-gather of the output" % (world * ch, seconds, world),
               "steps": steps, "warmup": warmup, "frames_per_step": frames, "allgather_bytes_per_rank": ch * F * HOP * 4, "api": self.api, "nranks": self.nranks,
               "slots_verified": legs["slots_verified"], "allgather_ms_alone": round(legs["gather_alone_ms"], 3)}
        for key, tag in (("compute_only_s", "compute_only"), ("then_gather_s", "compute_then_allgather"), ("overlapped_s", "overlapped")):
            t = legs[key] / steps
            out[tag] = {"ms_per_step": round(t * 1e3, 4), "frames_per_s": round(frames / t, 1)}
        del audio, pv, ws
        torch.cuda.empty_cache()
        return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    WINDOW, HOP, DFT = args.window, args.hop, args.dft
    BINS = DFT // 2 + 1
    # algorithmic bytes per PV frame -- analysis reads hop*4 and writes bins*8, synthesis mirrors it
    BYTES_ANALYSIS = BYTES_SYNTHESIS = HOP * 4 + BINS * 8      # 10 248 (dft 2048) / 18 440 (dft 4096, hop 512)
    BYTES_ROUNDTRIP = BYTES_ANALYSIS + BYTES_SYNTHESIS
    ch, n = args.channels, int(args.seconds * SR)
    parallelism = "channel-shard x%d" % world

    if args.plan_only:
        # the launcher, the rendezvous and the sharding arithmetic without a GPU (gloo): tests/test_bench_launcher.py
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        F = n // HOP + 1
        mine = {"rank": rank, "channels": [rank * ch, rank * ch + ch], "frames": ch * F}
        plans = [mine]
        if world > 1:
            plans = [None] * world
            dist.all_gather_object(plans, mine)
            dist.barrier()
        if rank == 0:
            from flan_amd import sharding
            n4 = int(args.config4_seconds * SR)
            F4 = n4 // HOP + 1
            c4 = {"workload": "BASELINE config 4: %d ch x %.0f s over %d GPUs, 8 ch per rank" % (8 * world, args.config4_seconds, world),
                  "channels_per_gpu": 8, "frames_per_step": world * 8 * F4,
                  "allgather": sharding.inplace_allgather_plan(world, 8, F4 * HOP),
                  "legs": ["compute_only", "compute_then_allgather", "overlapped"]} if world > 1 and not args.no_config4 else None
            print(json.dumps({"plan_only": True, "metric": baseline_metric(), "n_gpus": world, "scaling": "weak",
                              "config": {"workload": "%d ch x %.0f s per GPU" % (ch, args.seconds), "parallelism": parallelism},
                              "total_channels": world * ch, "frames_per_step": sum(p["frames"] for p in plans), "ranks": plans,
                              "allgather": sharding.inplace_allgather_plan(world, ch, F * HOP) if world > 1 else None, "config4": c4}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    import torch
    import flan_amd as fa
    from flan_amd import sharding

    # FLAN_BENCH_SHARE_GPU=1 rehearses the multi-rank path on ONE GPU: every rank uses cuda:0, the control plane runs over gloo and the
    # RCCL gather is left out (RCCL refuses two ranks on one device); numbers from it say nothing about scaling
    share_gpu = os.environ.get("FLAN_BENCH_SHARE_GPU") == "1"
    distributed = world > 1 or os.environ.get("FLAN_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: the RCCL path with one rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback path)")
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    fa.check(fa.lib.flanhip_set_device(dev_index))
    for kv in filter(None, args.kernel_variant.split(",")):
        which, variant = (int(v) for v in kv.split("="))
        fa.lib.flanhip_debug_option(which, variant)
    dev = torch.device("cuda", dev_index)
    dist = None
    if distributed:
        import torch.distributed as dist
        # rank 0 prints exactly one JSON line on stdout: RCCL's own messages go to stderr
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        if "MASTER_PORT" not in os.environ:                  # FLAN_BENCH_FORCE_DIST=1 outside a launcher: a one-rank job of its own
            os.environ["MASTER_PORT"] = str(free_port())
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # (a collective one rank never enters ends the job after three minutes, not after the backend's default ten)
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=180))
    ctl_dev = torch.device("cpu") if share_gpu else dev      # where the control-plane tensors (max over ranks) live

    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    ar = SR / HOP
    frames_per_step = ch * F
    stream = torch.cuda.current_stream().cuda_stream

    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234 + rank, ctypes.c_void_p(stream)))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, WINDOW), dtype=torch.uint8, device=dev)
    nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)

    # the round trip is fused by default: the analysis kernel leaves synthesis' pre-pass (per-chain phase sums) in the workspace
    def analyze(a=audio, p=pv, w=ws, c=ch, s=stream):
        if args.unfused:
            fa.analyze_dev(a, c, n, SR, WINDOW, HOP, DFT, p, s)
        else:
            fa.analyze_dev_fused(a, c, n, SR, WINDOW, HOP, DFT, p, w, s)

    def synthesize(p=pv, o=out, w=ws, c=ch, s=stream, stages=0xF):
        # stages (a per-call argument): which of the call's kernels are launched -- 1 k_phase_sums, 2 k_phase_scan, 4 k_synthesize, 8 k_ola_fixup;
        # the per-kernel timing below brackets one at a time, every other call launches all of them
        fa.synthesize_dev_stages(p, c, F, BINS, SR, ar, WINDOW, o, w, nan_flag, 0 if args.unfused else 1, stages, s)

    def step():
        analyze()
        synthesize()

    def sync_all():
        if distributed:
            # (the device drained first: the library's own communicator (flanhip_allgather_audio, side stream) and the process group's never have
            # collectives in flight at the same time -- two communicators' kernels waiting for each other's peers is the one way this job could hang)
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    # device warm-up (not the W warm-up steps of the contract, which follow): a GPU that has been idle needs tens of milliseconds of
    # load before its clocks settle; without this a short run measures the ramp, not the kernels
    # the cold figure first (what `--steps 5 --warmup 1 --preroll-ms 0` measures on a just-woken GPU): reported beside the steady-state value,
    # never as it
    cold = None
    if args.preroll_ms > 0 and not distributed:
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        cold_s = (time.perf_counter() - t0) / 5
        cold = {"steps": 5, "warmup": 1, "ms_per_step": round(cold_s * 1e3, 4), "frames_per_s": round(frames_per_step / cold_s, 1)}
    preroll_steps = 0
    if args.preroll_ms > 0:
        t_end = time.perf_counter() + args.preroll_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            preroll_steps += 10
    # ---- N > 1: the north star's path ends with ONE all-gather that reassembles the output, so the timed step of a multi-GPU job includes it:
    # every step is analysis -> synthesis straight into this rank's slot of the final channel-major buffer -> ONE in-place ncclAllGather of that buffer,
    # issued through the library's own C ABI (flanhip_comm_init / flanhip_allgather_audio, collective.hip) on a side stream, so that batch i's
    # gather travels under batch i + 1's compute (two final buffers in turn).  `value` is that; `value_compute_only` (>= 6x at N = 8 by construction,
    # channels being independent) and `value_compute_then_gather` (the same collective, not overlapped) stand beside it.  The communicator is set up
    # BEFORE anything is timed and the ranks agree (one all-reduce) that all of them have theirs; if the C-ABI binding cannot be had the same steps run
    # with torch.distributed's all_gather_into_tensor and the line says so.  Past that point nothing is caught: an error inside a collective ends the
    # rank, and the launcher ends the job -- a failed job, not a hung one.
    use_gather = distributed and not args.no_gather and not share_gpu
    gather = GatherLegs(fa, torch, dist, sharding, rank, world, dev, ctl_dev) if use_gather else None

    def timed_region(step_fn, finish_fn, steps=None, warmup=None):
        """W untimed steps, then exactly K steps between barriers + device synchronisation; the slowest rank's time"""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        for _ in range(warmup):
            step_fn()
        finish_fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        finish_fn()
        sync_all()
        dt = time.perf_counter() - t0
        return sharding.max_over_ranks(dist, dt, ctl_dev) if distributed else dt

    elapsed_compute = None
    legs = None
    if use_gather:
        # the headline shape with the gather: the K timed steps of the contract are the overlapped ones
        legs = gather.legs(audio, pv, ws, ch, n, WINDOW, HOP, DFT, args.steps, args.warmup, timed_region, unfused=args.unfused)
        elapsed, elapsed_compute = legs["overlapped_s"], legs["compute_only_s"]
    else:
        elapsed = timed_region(step, lambda: None)
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * frames_per_step * args.steps / elapsed
    value_compute_only = world * frames_per_step * args.steps / elapsed_compute if elapsed_compute else None
    # "hipEvent-timed, median of >= 20 runs" -- the same step, each one bracketed by events on the launch stream, reported
    # beside the contract's barrier-to-barrier mean over K steps
    n_med = max(20, min(args.steps, 200))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_med)]
    for e0, e1 in evs:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    per = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    step_ms_median = per[len(per) // 2]

    # per-kernel timing with events on the launch stream (rank 0 only), inside the same analyse -> synthesise sequence as the
    # timed region: events bracket k_analyze, the pre-pass (k_phase_sums unless fused, k_phase_scan), k_synthesize and
    # k_ola_fixup of every step (the `stages` argument of flanhip_synthesize_dev_stages selects which synthesis kernels a call launches)
    extra = {}
    roofline = None
    if rank == 0:
        reps = max(5, args.steps)
        acc = [0.0, 0.0, 0.0, 0.0]
        for _ in range(reps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            ev[0].record()
            analyze()
            ev[1].record()
            for i, mask in enumerate((1 | 2, 4, 8)):
                synthesize(stages=mask)
                ev[2 + i].record()
            torch.cuda.synchronize()
            for i in range(4):
                acc[i] += ev[i].elapsed_time(ev[i + 1]) / reps
        t_an, t_pre, t_sy_main, t_fix = acc
        extra["kernel_ms"] = {"k_analyze": round(t_an, 4), "prepass": round(t_pre, 4), "k_synthesize": round(t_sy_main, 4),
                              "k_ola_fixup": round(t_fix, 4), "fused": not args.unfused,
                              "note": "stage by stage for the events (flanhip_synthesize_dev_stages); the timed step itself is two dispatches where the synthesis "
                                      "kernel adds the chains' overlaps (dft 2048 / 1024 / 512, chains long enough): k_ola_fixup is then not launched"}
        if t_an >= t_sy_main:
            kname, tk, b = "k_analyze", t_an, BYTES_ANALYSIS
        else:
            kname, tk, b = "k_synthesize", t_sy_main, BYTES_SYNTHESIS
        achieved = frames_per_step * b / (tk * 1e-3) / 1e9
        # HBM bytes per launch of that kernel: not measurable from inside this process (PMC counters need rocprofv3), so the figure comes
        # from the committed PMC profile of this same workload and build (tools/scripts/profile_bench.sh -> profiles/r06_hbm_traffic.json:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, with the unit and gfx950 corrections the file explains) and is
        # labelled with its source; null for any other workload shape
        # the profile carries the hash of the kernel sources it was taken on (flan_amd/build.py: kernel_source_hash): a profile of other
        # kernels than the ones running is not quoted (traffic null, the reason in traffic_source)
        traffic, traffic_source = None, None
        try:
            if ch == 8 and abs(args.seconds - 60.0) < 1e-9 and DFT == 2048 and HOP == 512 and WINDOW == 2048:
                from flan_amd.build import kernel_source_hash
                path = os.path.join(ROOT, "profiles", "r06_hbm_traffic.json")
                with open(path) as fh:
                    prof = json.load(fh)
                if prof.get("kernel_source_hash") == kernel_source_hash():
                    traffic = prof[kname]["traffic_bytes"]
                    traffic_source = "profiles/r06_hbm_traffic.json"
                else:
                    traffic_source = "profiles/r06_hbm_traffic.json was taken on other kernel sources (%s, now %s): not quoted" % (str(prof.get("kernel_source_hash"))[:12], kernel_source_hash()[:12])
        except Exception as e:
            traffic, traffic_source = None, "no traffic profile: " + repr(e)
        # what a plain device-to-device copy of the same number of bytes reaches on this box (quote the measured
        # copy rate beside the 8 TB/s spec); read + written bytes, like the algorithmic figure
        # (round 6: the library's own 16-bytes-per-lane streaming copy, flanhip_copy_dev -- torch's copy_ measured 5.4 TB/s where this chip's
        # float4 copy reaches ~6.3, MI355X_MICROARCH.md; quoting the kernels against the slower one flattered them)
        copy_gbs = None
        try:
            cnt = (frames_per_step * b // 8) & ~3                  # floats: half the bytes read, half written
            src = pv.view(-1)[:cnt]
            dst = torch.empty_like(src)
            fa.check(fa.lib.flanhip_copy_dev(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), cnt, ctypes.c_void_p(stream)))
            best = None
            for _ in range(3):
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(10):
                    fa.check(fa.lib.flanhip_copy_dev(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), cnt, ctypes.c_void_p(stream)))
                c1.record()
                torch.cuda.synchronize()
                t = c0.elapsed_time(c1) / 10
                best = t if best is None else min(best, t)
            copy_gbs = round(2 * cnt * 4 / (best * 1e-3) / 1e9, 1)
            del dst
        except Exception:
            copy_gbs = None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "launch_ms": round(tk, 4), "algorithmic_bytes_per_launch": frames_per_step * b,
                    "copy_peak_measured": copy_gbs, "copy_peak_kind": "flanhip_copy_dev: 16 B per lane, grid-stride, the same bytes read + written"}
        # The bound that actually holds these kernels is the vector ALU, not HBM: the launch's VALU instruction mix, priced per
        # class with the measured issue costs, against the launch's own cycles.  PMC counters cannot be read from inside the process: the mix is
        # the committed profile's (profiles/r06_valu_roofline.json, tools/make_valu_roofline.py), quoted only while the kernel sources' hash
        # matches; the duration is the one measured live above, the clock the one the profile measured (GRBM_GUI_ACTIVE)
        roofline_valu = None
        try:
            if ch == 8 and abs(args.seconds - 60.0) < 1e-9 and DFT == 2048 and HOP == 512 and WINDOW == 2048 and not args.unfused:
                from flan_amd.build import kernel_source_hash
                with open(os.path.join(ROOT, "profiles", "r06_valu_roofline.json")) as fh:
                    vp_ = json.load(fh)
                if vp_.get("kernel_source_hash") == kernel_source_hash():
                    k_ = vp_[kname]
                    clock = k_.get("clock_ghz_in_profile") or 2.33
                    cycles = tk * 1e-3 * clock * 1e9
                    roofline_valu = {"bound": "valu", "kernel": kname, "priced_simd_cycles": k_["priced_simd_cycles_per_launch"], "launch_cycles": int(cycles),
                                     "frac": round(k_["priced_simd_cycles_per_launch"] / cycles, 4), "clock_ghz": clock, "insts_valu": k_["insts_valu"],
                                     "full_rate_floor_frac": round(k_["full_rate_floor_cycles"] / cycles, 4),
                                     "source": "profiles/r06_valu_roofline.json (SQ_INSTS_VALU_* per class x profiles/r03_a_issue_model.txt prices; LDS, VMEM and scalar issue not included)"}
                else:
                    roofline_valu = {"frac": None, "source": "profiles/r06_valu_roofline.json was taken on other kernel sources: not quoted"}
        except Exception as e:
            roofline_valu = {"frac": None, "source": "no instruction-mix profile: " + repr(e)}
        extra["roofline_valu"] = roofline_valu
        extra["roundtrip_hbm"] = {"achieved_GBs": round(frames_per_step * BYTES_ROUNDTRIP / (ms_per_step * 1e-3) / 1e9, 1),
                                  "frac_of_8TBs": round(frames_per_step * BYTES_ROUNDTRIP / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- the gather legs for the record, and BASELINE config 4's shape (64 ch x 10 min over 8 GPUs = 8 ch x 600 s per rank) through the same steps ----
    if use_gather:
        t_compute, t_then, t_over = (legs[k] / args.steps for k in ("compute_only_s", "then_gather_s", "overlapped_s"))
        extra["allgather"] = {"nranks": gather.nranks, "api": gather.api, "api_error": gather.err, "ms_alone": round(legs["gather_alone_ms"], 3),
                              "bytes_per_rank": ch * F * HOP * 4, "slots_verified": legs["slots_verified"],
                              "frames_per_s_compute_only": round(value_compute_only, 1),
                              "frames_per_s_compute_then_gather": round(world * frames_per_step / t_then, 1),
                              "frames_per_s_gather_overlapped": round(value, 1),
                              "overlapped_step_ms": round(t_over * 1e3, 4), "compute_then_gather_step_ms": round(t_then * 1e3, 4), "compute_only_step_ms": round(t_compute * 1e3, 4),
                              "overlap": "ONE in-place all-gather of the final [rank][channel][sample] buffer per step, on a side stream under the next step's compute (two buffers in turn)"}
        if not args.no_config4 and DFT == 2048 and HOP == 512 and WINDOW == 2048:
            del audio, pv, out, ws
            torch.cuda.empty_cache()
            try:
                extra["config4"] = gather.config4(args.config4_seconds, timed_region)
            except Exception as e:                    # (an allocation that does not fit, say: the headline line is not lost to it; errors INSIDE a collective are not caught)
                if "out of memory" not in repr(e).lower():
                    raise
                extra["config4"] = {"error": repr(e)}

    # the host-buffer C ABI (flanhip_analyze / flanhip_synthesize): upload, kernels, download -- for DESIGN.md, never the metric
    if rank == 0 and world == 1 and not use_gather and not args.no_pcie and ( args.pcie or ( ch * n <= 8 * 60 * 48000 and DFT <= 4096 ) ):
        import numpy as np
        x_host = audio.cpu().numpy()
        fa.analyze(x_host[:1, :48000], SR, WINDOW, HOP, DFT)
        t0 = time.perf_counter()
        pv_host = fa.analyze(x_host, SR, WINDOW, HOP, DFT)
        t1 = time.perf_counter()
        out_host, _ = fa.synthesize(pv_host, SR, np.float32(SR) / np.float32(HOP), WINDOW)
        t2 = time.perf_counter()
        extra["pcie_inclusive"] = {"analyze_ms": round((t1 - t0) * 1e3, 2), "synthesize_ms": round((t2 - t1) * 1e3, 2),
                                   "frames_per_s": round(frames_per_step / (t2 - t0), 1),
                                   "bytes_moved": int(x_host.nbytes + 2 * pv_host.nbytes + out_host.nbytes),
                                   "note": "pageable numpy buffers through flanhip_analyze + flanhip_synthesize (malloc, H2D, kernels, D2H)"}
        del pv_host, out_host

    # ---- the other BASELINE configurations, timed the same way (device resident, events, after the same warm-up) ----
    if rank == 0 and world == 1 and not args.no_configs and not use_gather:
        del pv, out, ws
        torch.cuda.empty_cache()
        try:
            extra["configs"] = other_configs(fa, torch, dev)
        except Exception as e:                       # never lose the headline line to a side measurement
            extra["configs"] = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:        # the CPU leg is reported at N=1 only; the other ranks would idle behind it
        cores = os.cpu_count() or 1
        sample_s = min(args.seconds, 60.0)           # bounded: at most 8 x 60 s = 45 008 frames (~4 s on one core, the same again on all)
        v1, frames1, dt1 = cpu_baseline(ch, sample_s, 1, WINDOW, HOP, DFT)
        share, fft_us = cpu_fft_share(frames1, DFT, dt1 / frames1)
        cpu = {"value": round(v1, 1), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d ch x %.0f s round trip (%d frames, %.1f s) by oracle/flan_oracle.cpp, 1 thread, fp64 radix-2 FFT (not FFTW)"
                         % (ch, sample_s, frames1, dt1),
               "fft_share": share, "fft_us_per_frame": fft_us,
               "fft_note": "r2c + c2r of the oracle's own FFT timed alone: this share of the oracle's time; FFTW would take ~3-5 us per transform"}
        threads = min(cores, ch)
        if threads > 1:
            vN, _, dtN = cpu_baseline(ch, sample_s, threads, WINDOW, HOP, DFT)
            cpu["all_cores"] = {"value": round(vN, 1), "cores": threads, "host_cores": cores, "seconds": round(dtN, 2)}

    if rank == 0:
        line = {
            "metric": baseline_metric(),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d ch x %.0f s x 48 kHz uniform noise per GPU, convert_to_PV(%d,%d,dft %d) -> convert_to_audio round trip"
                                   % (ch, args.seconds, WINDOW, HOP, DFT),
                       "channels_per_gpu": ch, "pv_frames_per_step_per_gpu": frames_per_step, "parallelism": parallelism},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if share_gpu:
            line["rehearsal"] = "FLAN_BENCH_SHARE_GPU=1: %d ranks on one GPU, gloo control plane, no RCCL gather" % world
        line["ms_per_step_median"] = round(step_ms_median, 4)          # median of >= 20 event-timed steps; ms_per_step: the K steps between barriers
        line["value_median"] = round(world * frames_per_step / (step_ms_median * 1e-3), 1) if world == 1 else None
        line["value_cold"] = cold["frames_per_s"] if cold else None       # a just-woken GPU, 5 steps after 1 (no clock pre-roll)
        ag = extra.get("allgather") or {}
        # N > 1: `value` is the step WITH the north star's output reassembly (the all-gather overlapped with the next batch's compute);
        # the compute-only figure and the not-overlapped one stand beside it.  N = 1: there is nothing to gather, `value` is the round trip
        line["value_includes_gather"] = bool(use_gather)
        # how the timed steps reassemble the output: one batch of point-to-point sends / receives per step straight into the final channel-major buffer
        # on a side stream (the same bytes over the same xGMI links as north_star's one in-place ncclAllGather, whose form is value_compute_then_gather)
        line["gather_kind"] = ("one in-place all-gather of the final [rank][channel][sample] buffer per step (%s), on a side stream under the next step's compute; "
                               "the same collective not overlapped is value_compute_then_gather" % gather.api) if use_gather else None
        line["value_compute_only"] = round(value_compute_only, 1) if value_compute_only else None
        line["value_gather_overlapped"] = ag.get("frames_per_s_gather_overlapped")
        line["value_compute_then_gather"] = ag.get("frames_per_s_compute_then_gather")
        line["rccl_nranks"] = ag.get("nranks")
        line["config4"] = extra.pop("config4", None)                       # N > 1: BASELINE config 4's shape per rank -- compute only, compute then all-gather, overlapped
        line["device_warmup"] = {"preroll_ms": args.preroll_ms, "preroll_steps": preroll_steps, "cold": cold}
        line.update(extra)
        print(json.dumps(line), flush=True)
    if gather is not None:
        gather.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def other_configs(fa, torch, dev):
    """BASELINE configs 3 and 5 and the reference API's default call, device resident, per-stage events.  Reported beside the headline
    ('configs'); never part of 'value'.  'ms' is the steady state (the median of five event-timed windows after ~60 ms of the same work, the
    headline's protocol), 'ms_cold' the first window."""
    lib = fa.lib
    vp = ctypes.c_void_p

    def P(t):
        return vp(t.data_ptr())

    def window(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    cold_ms = {}

    def timed(fn, reps=10, tag=None):
        """Steady state, like the headline: the first window (3 calls in, on a GPU that has just idled through allocations) is kept as the
        cold figure; then ~60 ms of the same work, and the median of five windows."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        cold = window(fn, reps)
        if tag is not None:
            cold_ms[tag] = round(cold, 4)
        for _ in range(int(60.0 / max(cold, 1e-3)) + 1):
            fn()
        torch.cuda.synchronize()
        w = sorted(window(fn, reps) for _ in range(5))
        return w[2]

    res = {}
    W, HOP, DFT = 2048, 512, 2048
    BINS = DFT // 2 + 1
    ch, n = 8, 60 * 48000
    F = int(lib.flanhip_num_pv_frames(n, HOP))
    ar = SR / HOP
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(P(audio), ch, n, 1234, None))

    # ---- config 2: stereo 60 s, convert_to_PV -> convert_to_audio round trip (fused, like the headline) with its kernels timed one by one
    c2 = 2
    pv2 = torch.empty((c2, F, BINS, 2), dtype=torch.float32, device=dev)
    out2 = torch.empty((c2, F * HOP), dtype=torch.float32, device=dev)
    ws2 = torch.empty(fa.synthesize_workspace_bytes(c2, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)

    def config2(stages=0xF):
        fa.analyze_dev_fused(audio, c2, n, SR, W, HOP, DFT, pv2, ws2, None)
        fa.synthesize_dev_stages(pv2, c2, F, BINS, SR, ar, W, out2, ws2, None, 1, stages, None)
    ms2 = timed(config2, tag="config2")
    k2 = [0.0] * 4
    reps2 = 20
    for _ in range(reps2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        fa.analyze_dev_fused(audio, c2, n, SR, W, HOP, DFT, pv2, ws2, None)
        ev[1].record()
        for i, mask in enumerate((1 | 2, 4, 8)):
            fa.synthesize_dev_stages(pv2, c2, F, BINS, SR, ar, W, out2, ws2, None, 1, mask, None)
            ev[2 + i].record()
        torch.cuda.synchronize()
        for i in range(4):
            k2[i] += ev[i].elapsed_time(ev[i + 1]) / reps2
    b_rt = 2 * (HOP * 4 + BINS * 8)
    res["config2_stereo_60s"] = {"workload": "2 ch x 60 s: convert_to_PV(2048,512,2048) -> convert_to_audio", "ms": round(ms2, 4), "ms_cold": cold_ms["config2"],
                                 "frames_per_s": round(c2 * F / (ms2 * 1e-3), 1), "algorithmic_GBs": round(c2 * F * b_rt / (ms2 * 1e-3) / 1e9, 1),
                                 "frac_of_8TBs": round(c2 * F * b_rt / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "kernel_ms": {"k_analyze": round(k2[0], 4), "scan (k_phase_scan2 over the group totals)": round(k2[1], 4),
                                               "k_synthesize": round(k2[2], 4), "k_ola_fixup": round(k2[3], 4),
                                               "note": "stage by stage; the step itself adds the overlaps inside the synthesis kernel"}}
    del pv2, out2, ws2

    # ---- config 3: 8 ch x 60 s -> convert_to_PV -> stretch( x2 ) -> convert_to_audio (PVModify.cpp:371-385, :307-362)
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    grid = torch.empty((F, BINS), dtype=torch.float32, device=dev)
    dmax = torch.empty(1, dtype=torch.float32, device=dev)
    Fo = 2 * F
    st = torch.empty((ch, Fo, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, Fo * HOP), dtype=torch.float32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, Fo, BINS, SR, ar, W), dtype=torch.uint8, device=dev)

    def config3():
        fa.check(lib.flanhip_analyze_dev(P(audio), ch, n, SR, W, HOP, DFT, P(pv), None))
        fa.check(lib.flanhip_stretch_map_const_dev(2.0, P(grid), F, BINS, SR, HOP, P(dmax), None))     # the constant factor's map in closed form (round 5; fill + scan before)
        fa.check(lib.flanhip_modify_time_dev_fused(P(pv), ch, F, BINS, SR, ar, P(grid), Fo, P(st), W, P(ws), None))
        fa.check(lib.flanhip_synthesize_dev_fused_checked(P(st), ch, Fo, BINS, SR, ar, W, P(out), P(ws), None, None))
    ms = timed(config3, tag="config3")
    bytes_per_input_frame = 10248 + 24600 + 2 * 10248                      # 55 344 B per input frame
    res["config3_stretch_x2"] = {"workload": "8 ch x 60 s: convert_to_PV(2048,512,2048) -> stretch(x2) -> convert_to_audio", "ms": round(ms, 4),
                                 "ms_cold": cold_ms["config3"], "input_frames_per_s": round(ch * F / (ms * 1e-3), 1),
                                 "algorithmic_GBs": round(ch * F * bytes_per_input_frame / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_8TBs": round(ch * F * bytes_per_input_frame / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del st, out, ws, grid

    # (BASELINE's order: config 3 before config 4's shard)
    # ---- config 4's per-GPU shard: 8 ch x 600 s (64 ch x 10 min over 8 GPUs), the same round trip on one GPU (a 3.7 GB PV)
    n4 = 600 * 48000
    F4 = int(lib.flanhip_num_pv_frames(n4, HOP))
    try:
        audio4 = torch.empty((ch, n4), dtype=torch.float32, device=dev)
        fa.check(lib.flanhip_noise_dev(P(audio4), ch, n4, 4321, None))
        pv4 = torch.empty((ch, F4, BINS, 2), dtype=torch.float32, device=dev)
        out4 = torch.empty((ch, F4 * HOP), dtype=torch.float32, device=dev)
        ws4 = torch.empty(fa.synthesize_workspace_bytes(ch, F4, BINS, SR, ar, W), dtype=torch.uint8, device=dev)

        def config4():
            fa.analyze_dev_fused(audio4, ch, n4, SR, W, HOP, DFT, pv4, ws4, None)
            fa.synthesize_dev_fused(pv4, ch, F4, BINS, SR, ar, W, out4, ws4, None, None)
        ms4 = timed(config4, 3, tag="config4")
        res["config4_shard_8ch_600s"] = {"workload": "8 ch x 600 s (one GPU's shard of 64 ch x 10 min over 8 GPUs): convert_to_PV(2048,512,2048) -> convert_to_audio",
                                         "ms": round(ms4, 4), "ms_cold": cold_ms["config4"], "frames_per_s": round(ch * F4 / (ms4 * 1e-3), 1),
                                         "algorithmic_GBs": round(ch * F4 * b_rt / (ms4 * 1e-3) / 1e9, 1),
                                         "frac_of_8TBs": round(ch * F4 * b_rt / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "pv_bytes": ch * F4 * BINS * 8}
        del audio4, pv4, out4, ws4
    except Exception as e:
        res["config4_shard_8ch_600s"] = {"error": repr(e)}
    torch.cuda.empty_cache()

    # ---- config 5: stereo 60 s at 96 kHz -> resample( 48000 ) -> convert_to_PV -> shape( f + 100 Hz ) -> convert_to_audio
    c5, n96, n48 = 2, 60 * 96000, 60 * 48000
    x96 = torch.empty((c5, n96), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(P(x96), c5, n96, 7, None))
    x48 = torch.empty((c5, n48), dtype=torch.float32, device=dev)
    F5 = int(lib.flanhip_num_pv_frames(n48, HOP))
    pv5 = torch.empty((c5, F5, BINS, 2), dtype=torch.float32, device=dev)
    sh5 = torch.empty_like(pv5)
    out5 = torch.empty((c5, F5 * HOP), dtype=torch.float32, device=dev)
    ws5 = torch.empty(fa.synthesize_workspace_bytes(c5, F5, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
    stages5 = {
        "resample_96k_to_48k": lambda: fa.check(lib.flanhip_resample_dev(P(x96), c5, n96, 96000.0, 48000.0, P(x48), None)),
        "convert_to_PV": lambda: fa.check(lib.flanhip_analyze_dev(P(x48), c5, n48, SR, W, HOP, DFT, P(pv5), None)),
        "shape_f_plus_100_and_convert_to_audio": lambda: (
            fa.check(lib.flanhip_shape_affine_dev_fused(P(pv5), c5, F5, BINS, SR, ar, 1.0, 0.0, 1.0, 100.0, P(sh5), W, P(ws5), None)),
            fa.check(lib.flanhip_synthesize_dev_fused(P(sh5), c5, F5, BINS, SR, ar, W, P(out5), P(ws5), None, None))),
    }
    st5 = {k: round(timed(v, 5, tag=k), 4) for k, v in stages5.items()}
    total5 = sum(st5.values())
    res["config5_resample_shape"] = {"workload": "2 ch x 60 s at 96 kHz: resample(48000) -> convert_to_PV(2048,512,2048) -> shape(f+100 Hz) -> convert_to_audio",
                                     "stage_ms": st5, "ms": round(total5, 4), "ms_cold": round(sum(cold_ms[k] for k in stages5), 4),
                                     "pv_frames_per_s": round(c5 * F5 / (total5 * 1e-3), 1)}
    del x96, x48, pv5, sh5, out5, ws5, pv

    # ---- the reference API's own defaults: convert_to_PV() = ( 2048, 128, 4096 ) (Audio.h:158-163), and ( 2048, 512, 4096 )
    # ... and ( 4096, 1024, 4096 ): window = dft, the plain STFT call at that size (the team kernels' one-buffer-set variants)
    # ... and the power-of-two sizes below the metric's: ( 1024, 256, 1024 ) -- the classic setting -- (pv_kernels_v3.h, round 5), ( 512, 128, 512 ) and ( 256, 64, 256 )
    # (pv_kernels_sub.h, round 6: several chains per wavefront)
    # ... and two sizes FFTW plans like any other (FFTHelper.cpp:16-26): ( 2048, 512, 2998 ), half the size 1499 a prime (Bluestein's chirp-z form,
    # pv_kernels_bs.h), and ( 4096, 1024, 32768 ) (residue pairs, pv_kernels_big.h: a 2.95 GB PV) -- both direct sums until round 5; ( 4096, 1024, 20000 ): the
    # residue pairs with a mixed-radix inner transform (20000 = 2 x 4 x 2500; round 6, direct sums before)
    for (hop, tag, Wd, dft) in ((128, "api_default_2048_128_4096", 2048, 4096), (512, "dft4096_hop512", 2048, 4096), (1024, "window4096_hop1024_dft4096", 4096, 4096),
                                (256, "dft1024_window1024_hop256", 1024, 1024), (128, "dft512_window512_hop128", 512, 512), (64, "dft256_window256_hop64", 256, 256),
                                (32, "dft128_window128_hop32", 128, 128),
                                # ... and dft 8192 / 16384: teams of four / eight wavefronts per chain (pv_kernels_team.h, round 6; before: the round-1 block kernels 0.92 ms, mixed radix 4.4 ms)
                                (2048, "dft8192_window8192_hop2048", 8192, 8192), (1024, "dft16384_window4096_hop1024", 4096, 16384),
                                (512, "dft2998_window2048_hop512_chirp_z", 2048, 2998), (1024, "dft32768_window4096_hop1024", 4096, 32768),
                                (1024, "dft20000_window4096_hop1024_mixed_radix", 4096, 20000),
                                # ... the reference API's default ratio (window = dft / 2, hop = window / 16) at dft 16384: the team kernels at half a step (round 6; it ran the
                                # DIRECT SUMS before: 786 ms), and a size with a large prime factor above 8192 (9998 = 2 x 4999): chirp-z in device memory (round 6; ~150 ms before)
                                (512, "dft16384_window8192_hop512_api_ratio", 8192, 16384), (2499, "dft9998_window9998_hop2499_chirp_z_in_memory", 9998, 9998)):
        bins = dft // 2 + 1
        Fd = int(lib.flanhip_num_pv_frames(n, hop))
        ard = SR / hop
        pvd = torch.empty((ch, Fd, bins, 2), dtype=torch.float32, device=dev)
        outd = torch.empty((ch, Fd * hop), dtype=torch.float32, device=dev)
        wsd = torch.empty(fa.synthesize_workspace_bytes(ch, Fd, bins, SR, ard, Wd), dtype=torch.uint8, device=dev)

        def rt():
            fa.analyze_dev_fused(audio, ch, n, SR, Wd, hop, dft, pvd, wsd, None)
            fa.synthesize_dev_fused(pvd, ch, Fd, bins, SR, ard, Wd, outd, wsd, None, None)
        msd = timed(rt, 5, tag=tag)
        b = 2 * (hop * 4 + bins * 8)
        res[tag] = {"workload": "8 ch x 60 s: convert_to_PV(%d,%d,%d) -> convert_to_audio" % (Wd, hop, dft), "ms": round(msd, 4), "ms_cold": cold_ms[tag],
                    "frames_per_s": round(ch * Fd / (msd * 1e-3), 1), "algorithmic_GBs": round(ch * Fd * b / (msd * 1e-3) / 1e9, 1),
                    "frac_of_8TBs": round(ch * Fd * b / (msd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del pvd, outd, wsd
    return res


if __name__ == "__main__":
    main()
