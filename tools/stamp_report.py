#!/usr/bin/env python3
"""Per-section cycle shares of the stamped v2 kernels (diagnostic build: tools/scripts/build_diag.sh stamps).

    FLAN_AMD_LIB=tools/ubench/libflanhip_stamps.so python tools/stamp_report.py --ana 4,2 [--fused]

Prints, per variant, the wave-cycles (s_memtime ticks) each section of the frame loop took, per frame and as a share.
The stamps' fences change the schedule: read shares, not totals."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES_ANA = ["wait samples + window", "pass 0 + transpose", "pass 1 + transpose", "pass 2", "mirror write + upper loads",
             "bins group 0", "bins group 1", "bins group 2", "bins group 3", "bin C/2 + sync", "-", "-"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ana", default="4")
    ap.add_argument("--fused", action="store_true")
    args = ap.parse_args()
    import torch
    import flan_amd as fa
    lib = ctypes.CDLL(fa.LIB_PATH)
    if not hasattr(lib, "flanhip_debug_read_stamps"):
        raise SystemExit("not a stamped build: set FLAN_AMD_LIB=tools/ubench/libflanhip_stamps.so")
    W, HOP, DFT, SR = 2048, 512, 2048, 48000.0
    BINS = DFT // 2 + 1
    dev = torch.device("cuda", 0)
    fa.check(fa.lib.flanhip_set_device(0))
    ch, n = 8, int(60 * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    stream = torch.cuda.current_stream().cuda_stream
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, ctypes.c_void_p(stream)))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    buf = (ctypes.c_ulonglong * 16)()
    for v in [int(x) for x in args.ana.split(",")]:
        fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, v)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, SR / HOP, W), dtype=torch.uint8, device=dev)
        run = (lambda: fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)) if args.fused else (lambda: fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv, stream))
        for _ in range(50):
            run()
        lib.flanhip_debug_read_stamps(buf)
        reps = 20
        for _ in range(reps):
            run()
        lib.flanhip_debug_read_stamps(buf)
        vals = list(buf)
        waves = vals[15]
        total = sum(vals[:12])
        frames = ch * F * reps
        print("analysis variant %d%s: %d wavefronts, %.0f ticks per frame (all sections)" % (v, " fused" if args.fused else "", waves // reps, total / max(frames, 1)))
        if vals[13]:
            print("   in-kernel clock: %.0f MHz (s_memtime ticks per 100 MHz s_memrealtime tick x 100); wave life %.1f us" % (100.0 * vals[12] / vals[13], vals[13] / max(waves, 1) / 100.0))
        for i, nm in enumerate(NAMES_ANA):
            if vals[i]:
                print("   %-28s %8.0f ticks/frame  %5.1f %%" % (nm, vals[i] / frames, 100.0 * vals[i] / total))
    fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, 0)


if __name__ == "__main__":
    main()
