#!/usr/bin/env python3
"""Does the run time of the dft 2048 kernels depend on the DATA (same instruction stream)?  A kernel that is held back by the chip's
power management runs faster on inputs that toggle fewer bits; one that is bound by issue slots or latency does not.

    python tools/data_dependence.py [--rounds 7] [--reps 40]

Times the fused analysis and synthesis launches (8 ch x 60 s) on: uniform noise (the bench input), a 440 Hz sine, a constant,
noise scaled by 2^-20 (same mantissas, other exponents) and noise with the 16 low mantissa bits cleared; interleaved rounds in one process."""
import argparse
import ctypes
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=40)
    args = ap.parse_args()
    import torch
    import flan_amd as fa
    W, HOP, DFT, SR = 2048, 512, 2048, 48000.0
    BINS = DFT // 2 + 1
    dev = torch.device("cuda", 0)
    fa.check(fa.lib.flanhip_set_device(0))
    ch, n = 8, int(60 * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    ar = SR / HOP
    stream = torch.cuda.current_stream().cuda_stream
    noise = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(noise.data_ptr()), ch, n, 1234, ctypes.c_void_p(stream)))
    t = torch.arange(n, dtype=torch.float64, device=dev)
    inputs = {
        "uniform noise": noise,
        "sine 440 Hz": (0.5 * torch.sin(2 * math.pi * 440.0 * t / SR)).float().repeat(ch, 1).contiguous(),
        "constant 0.25": torch.full((ch, n), 0.25, dtype=torch.float32, device=dev),
        "noise x 2^-20": (noise * 2.0 ** -20).contiguous(),
        "noise, 16 low mantissa bits cleared": (noise.view(torch.int32) & ~0xFFFF).view(torch.float32).contiguous(),
    }
    pvs = {k: torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev) for k in inputs}
    wss = {k: torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev) for k in inputs}
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def time_call(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ana = {k: (lambda k=k: fa.analyze_dev_fused(inputs[k], ch, n, SR, W, HOP, DFT, pvs[k], wss[k], stream)) for k in inputs}
    # the main synthesis kernel alone (stages = 4)
    syn = {k: (lambda k=k: fa.synthesize_dev_stages(pvs[k], ch, F, BINS, SR, ar, W, out, wss[k], nan_flag, 1, 4, stream)) for k in inputs}
    for k in inputs:
        ana[k]()
    for _ in range(300):
        ana["uniform noise"]()
    torch.cuda.synchronize()
    res = {k: {"analysis_ms": [], "synthesis_ms": []} for k in inputs}
    for _ in range(args.rounds):
        for k in inputs:
            ana[k]()
            res[k]["analysis_ms"].append(time_call(ana[k], args.reps))
        for k in inputs:
            syn[k]()
            res[k]["synthesis_ms"].append(time_call(syn[k], args.reps))
    for k in inputs:
        a, s = sorted(res[k]["analysis_ms"]), sorted(res[k]["synthesis_ms"])
        print("%-40s analysis median %.4f min %.4f ms   synthesis median %.4f min %.4f ms" % (k, a[len(a) // 2], a[0], s[len(s) // 2], s[0]))


if __name__ == "__main__":
    main()
