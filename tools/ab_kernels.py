#!/usr/bin/env python3
"""A/B timing of kernel variants in ONE process, interleaved rounds (flanhip_debug_option).

    python tools/ab_kernels.py --ana 0,1,2,3,4 [--fused] [--syn 0,1] [--rounds 7] [--reps 20]

Prints per variant: median / min ms of the analysis (or synthesis) launch for 8 ch x 60 s, the algorithmic HBM fraction,
and how its output differs from variant 0's on the same input (bit-identical f? relative difference of m).
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ana", default="0,1")
    ap.add_argument("--syn", default="")
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import flan_amd as fa
    W, HOP, DFT, SR = 2048, 512, 2048, 48000.0
    BINS = DFT // 2 + 1
    dev = torch.device("cuda", 0)
    fa.check(fa.lib.flanhip_set_device(0))
    ch, n = args.channels, int(args.seconds * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    ar = SR / HOP
    stream = torch.cuda.current_stream().cuda_stream
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, ctypes.c_void_p(stream)))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    bytes_per_launch = ch * F * (HOP * 4 + BINS * 8)
    res = {"frames": ch * F, "analysis": {}, "synthesis": {}}

    def ws_for():
        return torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)

    def time_call(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ana = [int(v) for v in args.ana.split(",") if v != ""]
    syn = [int(v) for v in args.syn.split(",") if v != ""]
    # ---- analysis variants
    if ana:
        ref = None
        runs = {}
        for v in ana:
            fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, v)
            ws = ws_for()
            pv.zero_()
            if args.fused:
                fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
            else:
                fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv, stream)
            torch.cuda.synchronize()
            cur = pv.clone()
            info = {}
            if ref is None:
                ref = cur
            else:
                fb = (cur[..., 1].view(torch.int32) != ref[..., 1].view(torch.int32))
                info["f_bits_differ"] = int(fb.sum().item())
                info["f_bits_differ_not_bin512"] = int(fb.sum().item() - fb[:, :, 512].sum().item())
                dm = (cur[..., 0].double() - ref[..., 0].double())
                info["m_rel_l2"] = float(torch.sqrt((dm * dm).sum() / (ref[..., 0].double() ** 2).sum()).item())
                info["m_max_rel"] = float((dm.abs() / ref[..., 0].double().abs().clamp_min(1e-30)).max().item())
            runs[v] = {"info": info, "ms": [], "ws": ws}
        # warm the clocks
        for _ in range(100):
            fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv, stream)
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v in ana:
                fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, v)
                ws = runs[v]["ws"]
                if args.fused:
                    fn = lambda: fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
                else:
                    fn = lambda: fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv, stream)
                fn()
                runs[v]["ms"].append(time_call(fn, args.reps))
        for v in ana:
            ms = sorted(runs[v]["ms"])
            med = ms[len(ms) // 2]
            res["analysis"][v] = {"median_ms": round(med, 4), "min_ms": round(ms[0], 4), "hbm_frac_algorithmic": round(bytes_per_launch / (med * 1e-3) / 8e12, 4), **runs[v]["info"]}
        fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, 0)
    # ---- synthesis variants (main kernel only: stage mask 4; the PV and the carries come from a fused analysis with variant 0 layout)
    if syn:
        ref = None
        runs = {}
        for v in syn:
            fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, v)
            ws = ws_for()
            fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv, stream)
            out.zero_()
            fa.synthesize_dev(pv, ch, F, BINS, SR, ar, W, out, ws, nan_flag, stream)
            torch.cuda.synchronize()
            cur = out.clone()
            info = {}
            if ref is None:
                ref = cur
            else:
                d = cur.double() - ref.double()
                info["rms_vs_variant0"] = float(torch.sqrt((d * d).mean()).item())
                info["max_abs_vs_variant0"] = float(d.abs().max().item())
                info["bits_differ"] = int((cur.view(torch.int32) != ref.view(torch.int32)).sum().item())
            runs[v] = {"info": info, "ms": [], "ws": ws}
        for _ in range(args.rounds):
            for v in syn:
                fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, v)
                ws = runs[v]["ws"]
                fa.synthesize_dev(pv, ch, F, BINS, SR, ar, W, out, ws, nan_flag, stream)   # carries for this layout
                fn = lambda: fa.synthesize_dev_stages(pv, ch, F, BINS, SR, ar, W, out, ws, nan_flag, 0, 4, stream)   # the main kernel alone
                fn()
                runs[v]["ms"].append(time_call(fn, args.reps))
        for v in syn:
            ms = sorted(runs[v]["ms"])
            med = ms[len(ms) // 2]
            res["synthesis"][v] = {"median_ms": round(med, 4), "min_ms": round(ms[0], 4), "hbm_frac_algorithmic": round(bytes_per_launch / (med * 1e-3) / 8e12, 4), **runs[v]["info"]}
        fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, 0)
    text = json.dumps(res, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
