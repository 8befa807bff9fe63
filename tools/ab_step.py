#!/usr/bin/env python3
"""Whole fused round trip (analysis + everything convert_to_audio launches) per kernel-variant pair, interleaved in one process.
    python tools/ab_step.py --pairs 0:0,0:2 [--channels 1 --seconds 480 --dft 2048 --hop 512]
(synthesis variant 2: the scan kernel instead of the carry prologue from group totals)"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", default="4:1,4:2,0:0")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--dft", type=int, default=2048)
    ap.add_argument("--hop", type=int, default=512)
    args = ap.parse_args()
    import torch
    import flan_amd as fa
    W, HOP, DFT, SR = 2048, args.hop, args.dft, 48000.0
    BINS = DFT // 2 + 1
    dev = torch.device("cuda", 0)
    fa.check(fa.lib.flanhip_set_device(0))
    ch, n = args.channels, int(args.seconds * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    ar = SR / HOP
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, None))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    pairs = [tuple(int(v) for v in p.split(":")) for p in args.pairs.split(",")]
    state = {}
    for (a, s) in pairs:
        fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, a); fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, s)
        state[(a, s)] = {"ws": torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev), "ms": []}

    def step(a, s):
        ws = state[(a, s)]["ws"]
        fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, None)
        fa.synthesize_dev_fused(pv, ch, F, BINS, SR, ar, W, out, ws, flag, None)
    ref = None
    for (a, s) in pairs:
        fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, a); fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, s)
        for _ in range(60):
            step(a, s)
        torch.cuda.synchronize()
        cur = out.clone()
        if ref is None:
            ref = cur
        else:
            d = (cur.double() - ref.double())
            state[(a, s)]["rms_vs_first"] = float(torch.sqrt((d * d).mean()).item())
    for _ in range(args.rounds):
        for (a, s) in pairs:
            fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, a); fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, s)
            step(a, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                step(a, s)
            e1.record(); torch.cuda.synchronize()
            state[(a, s)]["ms"].append(e0.elapsed_time(e1) / args.reps)
    for k, v in state.items():
        ms = sorted(v["ms"])
        print("ana %d syn %d : median %.4f ms  min %.4f ms  (%.1f M frames/s)  rms vs first %s" % (k[0], k[1], ms[len(ms) // 2], ms[0], ch * F / ms[len(ms) // 2] / 1e3, v.get("rms_vs_first")))
    fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, 0); fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, 0)


if __name__ == "__main__":
    main()
