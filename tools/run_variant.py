#!/usr/bin/env python3
"""Launch one analysis / synthesis kernel variant a few times (for rocprofv3 --pmc runs: tools/scripts/pmc_variants.sh).
    python tools/run_variant.py --ana 4 [--fused] [--syn 1] [--reps 5]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ana", type=int, default=0)
    ap.add_argument("--syn", type=int, default=0)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
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
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, ctypes.c_void_p(stream)))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, args.ana)
    fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, args.syn)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
    for _ in range(args.reps):
        if args.fused:
            fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
            fa.synthesize_dev_fused(pv, ch, F, BINS, SR, ar, W, out, ws, nan_flag, stream)
        else:
            fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv, stream)
            fa.synthesize_dev(pv, ch, F, BINS, SR, ar, W, out, ws, nan_flag, stream)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
