#!/usr/bin/env python3
"""bench.py -- PV analysis+resynthesis frames/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic audio that is already resident in HBM:
    Audio::convert_to_PV(2048, 512, dft 2048)  ->  PV::convert_to_audio        (flanhip_analyze_dev + flanhip_synthesize_dev)
Workload at N=1: 8 channels x 60 s x 48 kHz uniform noise (the configuration BASELINE.json's north_star quotes its
targets on: "60 s x 8-ch 48 kHz convertToPV->convertToAudio round-trip at 1 GPU").  With --gpus N every rank owns its
own 8 channels (channels are independent: AudioPV.cpp:41,44,108,111), so the job is 8N channels, "scaling": "weak",
with no collective inside the timed region.  The RCCL all-gather that reassembles the output buffer
(float[8N][frames], channel-major so the gathered buffer IS the final layout) is timed separately and reported in
"allgather" -- it is not part of the PV frames/s metric.

PyTorch here is plumbing only: device buffers, the stream, events and torch.distributed.  The kernels are the HIP
library flan_amd/libflanhip.so called through the C ABI (include/flanhip.h).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WINDOW, HOP, DFT, SR = 2048, 512, 2048, 48000.0
BINS = DFT // 2 + 1
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
# algorithmic bytes per PV frame at dft 2048 -- analysis reads hop*4 and writes bins*8, synthesis mirrors it
BYTES_ANALYSIS = HOP * 4 + BINS * 8         # 10 248
BYTES_SYNTHESIS = BINS * 8 + HOP * 4        # 10 248
BYTES_ROUNDTRIP = BYTES_ANALYSIS + BYTES_SYNTHESIS   # 20 496


def baseline_metric():
    """the metric string exactly as BASELINE.json names it"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except Exception:
        return "PV analysis+resynthesis frames/sec (2048-win, hop 512, 48 kHz) at 1/2/4/8 GPU"


def cpu_baseline(channels, seconds, threads):
    """The CPU oracle (oracle/flan_oracle.cpp, a port of the reference path) on the host cores; checker code used as the
    reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    n = int(seconds * SR)
    x = O.noise(channels, n, seed=1234)

    def one(c):
        pv = O.analyze(x[c:c + 1], SR, WINDOW, HOP, DFT)
        out, _ = O.synthesize(pv, SR, SR / HOP, WINDOW)
        return pv.shape[1]

    t0 = time.perf_counter()
    if threads <= 1:
        frames = sum(one(c) for c in range(channels))
    else:
        with ThreadPoolExecutor(threads) as ex:       # ctypes releases the GIL; one channel per task
            frames = sum(ex.map(one, range(channels)))
    dt = time.perf_counter() - t0
    return frames / dt, frames, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=8, help="channels per GPU")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="run synthesis' pre-pass as its own kernel instead of inside analysis")
    ap.add_argument("--preroll-ms", type=float, default=80.0,
                    help="untimed device warm-up before the W warm-up steps: the same steps run for this long so that the GPU's clocks "
                         "have settled (measured: a launch takes 0.204 ms on a just-woken MI355X, 0.166 ms after ~40 ms of load); 0 disables")
    ap.add_argument("--pcie", action="store_true", help="also time the host-buffer entry points (PCIe inclusive; reported in 'pcie_inclusive', never in 'value')")
    ap.add_argument("--window", type=int, default=2048, help="2048: the BASELINE metric")
    ap.add_argument("--hop", type=int, default=512, help="512: the BASELINE metric; 128 with --dft 4096 is the reference API's default call")
    ap.add_argument("--dft", type=int, default=2048,
                    help="2048: the primary measurement; 4096: the literal convert_to_PV(2048,512) default of the reference API (SURVEY 8)")
    args = ap.parse_args()
    global DFT, HOP, WINDOW, BINS, BYTES_ANALYSIS, BYTES_SYNTHESIS, BYTES_ROUNDTRIP
    DFT, HOP, WINDOW = args.dft, args.hop, args.window
    BINS = DFT // 2 + 1
    BYTES_ANALYSIS = BYTES_SYNTHESIS = HOP * 4 + BINS * 8      # 10 248 (dft 2048) / 18 440 (dft 4096)
    BYTES_ROUNDTRIP = BYTES_ANALYSIS + BYTES_SYNTHESIS

    import torch
    import flan_amd as fa
    from flan_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    distributed = world > 1 or os.environ.get("FLAN_BENCH_FORCE_DIST") == "1"   # the env knob rehearses the RCCL path on one GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback path)")
    torch.cuda.set_device(local_rank)
    fa.check(fa.lib.flanhip_set_device(local_rank))
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rank 0 prints exactly one JSON line on stdout: RCCL's own messages (its version banner, warnings such as "Could not read
        # node #") go to stderr
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        dist.init_process_group("nccl", device_id=dev)

    ch, n = args.channels, int(args.seconds * SR)
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
    def analyze():
        if args.unfused:
            fa.analyze_dev(audio, ch, n, SR, WINDOW, HOP, DFT, pv, stream)
        else:
            fa.analyze_dev_fused(audio, ch, n, SR, WINDOW, HOP, DFT, pv, ws, stream)

    def synthesize():
        if args.unfused:
            fa.synthesize_dev(pv, ch, F, BINS, SR, ar, WINDOW, out, ws, nan_flag, stream)
        else:
            fa.synthesize_dev_fused(pv, ch, F, BINS, SR, ar, WINDOW, out, ws, nan_flag, stream)

    def step():
        analyze()
        synthesize()

    def sync_all():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # device warm-up (not the W warm-up steps of the contract, which follow): a GPU that has been idle needs tens of milliseconds of
    # load before its clocks settle; without this a short run measures the ramp, not the kernels
    preroll_steps = 0
    if args.preroll_ms > 0:
        t_end = time.perf_counter() + args.preroll_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            preroll_steps += 10
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if distributed:
        elapsed = sharding.max_over_ranks(dist, elapsed, dev)
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * frames_per_step * args.steps / elapsed

    # per-kernel timing with events on the launch stream (rank 0 only), inside the same analyse -> synthesise sequence as the
    # timed region: events bracket k_analyze, the pre-pass (k_phase_sums unless fused, k_phase_scan), k_synthesize and
    # k_ola_fixup of every step (flanhip_debug_synth_stages selects which synthesis kernels a call launches)
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
                fa.lib.flanhip_debug_synth_stages(mask)
                synthesize()
                ev[2 + i].record()
            torch.cuda.synchronize()
            for i in range(4):
                acc[i] += ev[i].elapsed_time(ev[i + 1]) / reps
        fa.lib.flanhip_debug_synth_stages(0xF)
        t_an, t_pre, t_sy_main, t_fix = acc
        extra["kernel_ms"] = {"k_analyze": round(t_an, 4), "prepass": round(t_pre, 4), "k_synthesize": round(t_sy_main, 4),
                              "k_ola_fixup": round(t_fix, 4), "fused": not args.unfused}
        if t_an >= t_sy_main:
            kname, tk, b = "k_analyze", t_an, BYTES_ANALYSIS
        else:
            kname, tk, b = "k_synthesize", t_sy_main, BYTES_SYNTHESIS
        achieved = frames_per_step * b / (tk * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed PMC profile of this same workload (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, corrected as profiles/r01_hbm_traffic.json explains); null for any other workload shape
        traffic = None
        try:
            if ch == 8 and abs(args.seconds - 60.0) < 1e-9 and DFT == 2048 and HOP == 512 and WINDOW == 2048:
                with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as fh:
                    traffic = json.load(fh)[kname]["traffic_bytes"]
        except Exception:
            traffic = None
        # what a plain device-to-device copy of the same number of bytes reaches on this box (quote the measured
        # copy rate beside the 8 TB/s spec); read + written bytes, like the algorithmic figure
        copy_gbs = None
        try:
            src = pv.view(-1)[: frames_per_step * b // 8]          # half the bytes read, half written
            dst = torch.empty_like(src)
            dst.copy_(src)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(10):
                dst.copy_(src)
            c1.record()
            torch.cuda.synchronize()
            copy_gbs = round(2 * src.numel() * 4 / (c0.elapsed_time(c1) / 10 * 1e-3) / 1e9, 1)
            del dst
        except Exception:
            copy_gbs = None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launch_ms": round(tk, 4), "algorithmic_bytes_per_launch": frames_per_step * b,
                    "copy_peak_measured": copy_gbs}
        extra["roundtrip_hbm"] = {"achieved_GBs": round(frames_per_step * BYTES_ROUNDTRIP / (ms_per_step * 1e-3) / 1e9, 1),
                                  "frac_of_8TBs": round(frames_per_step * BYTES_ROUNDTRIP / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # the output-reassembly all-gather of the north star, outside the metric
    if distributed and not args.no_gather:
        for _ in range(2):
            gathered = sharding.gather_output(dist, out, world)
        sync_all()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            gathered = sharding.gather_output(dist, out, world)
        sync_all()
        tg = sharding.max_over_ranks(dist, (time.perf_counter() - t0) / reps, dev)
        assert gathered.shape == (world * ch, F * HOP)
        extra["allgather"] = {"ms": round(tg * 1e3, 3), "bytes_per_rank": out.numel() * 4,
                              "frames_per_s_with_gather": round(world * frames_per_step / (elapsed / args.steps + tg), 1)}

    # the host-buffer C ABI (flanhip_analyze / flanhip_synthesize): upload, kernels, download -- for DESIGN.md, never the metric
    if rank == 0 and args.pcie:
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

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:        # the CPU leg is reported at N=1 only; the other ranks would idle behind it
        cores = os.cpu_count() or 1
        v1, frames1, dt1 = cpu_baseline(ch, args.seconds, 1)
        cpu = {"value": round(v1, 1), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d ch x %.0f s round trip (%d frames, %.1f s) by oracle/flan_oracle.cpp, 1 thread, fp64 radix-2 FFT (not FFTW)"
                         % (ch, args.seconds, frames1, dt1)}
        threads = min(cores, ch)
        if threads > 1:
            vN, _, dtN = cpu_baseline(ch, args.seconds, threads)
            cpu["all_cores"] = {"value": round(vN, 1), "cores": threads, "host_cores": cores, "seconds": round(dtN, 2)}

    if rank == 0:
        line = {
            "metric": baseline_metric(),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d ch x %.0f s x 48 kHz uniform noise per GPU, convert_to_PV(%d,%d,dft %d) -> convert_to_audio round trip"
                                   % (ch, args.seconds, WINDOW, HOP, DFT),
                       "channels_per_gpu": ch, "pv_frames_per_step_per_gpu": frames_per_step, "parallelism": "channel-shard x%d" % world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line["device_warmup"] = {"preroll_ms": args.preroll_ms, "preroll_steps": preroll_steps}
        line.update(extra)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
