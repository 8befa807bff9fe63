#!/usr/bin/env python3
"""Run the phase-ablated variants of the dft 2048 kernels once each, for a PMC pass (tools/scripts/valu_budget.sh): the difference in
SQ_INSTS_VALU between the whole kernel and the kernel without a phase is what that phase issues per launch.

    FLAN_AMD_LIB=tools/ubench/libflanhip_ablations.so python tools/valu_budget.py          (diagnostic build: build_diag.sh ablations)
    python tools/valu_budget.py --report gpurun_out/valu_budget_raw.txt                    (turn the PMC summary into the table)"""
import argparse
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ANA = [4, 101, 102, 104, 108, 116, 132, 163]     # whole; without: FFT passes, atan2+magnitude, wrap arithmetic, MF stores, sample loads, mirror read; skeleton
SYN = [1, 108]                                   # whole; without the transform (fft + overlap-add): what the bins cost
FRAMES = 45008


def run():
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
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
    for a in ANA:
        fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, a)
        fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, 1)
        for _ in range(3):
            fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
        torch.cuda.synchronize()
    fa.lib.flanhip_debug_option(fa.DEBUG_ANA_VARIANT, 4)
    fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)
    for s in SYN:
        fa.lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, s)
        for _ in range(3):
            fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, stream)      # the synthesis consumes the producer note
            fa.synthesize_dev_fused(pv, ch, F, BINS, SR, ar, W, out, ws, nan_flag, stream)
        torch.cuda.synchronize()


def report(path):
    rows, name = {}, None
    for line in open(path):
        m = re.match(r"void flanhip::k_(analyze|synthesize)_v2<8, (?:true, 8|4), (\d+)>", line)
        if m:
            name = (m.group(1), int(m.group(2)))
            continue
        if name and "=" in line:
            rows.setdefault(name, {}).update({k: float(v) for k, v in (kv.split("=") for kv in line.split())})
            name = None
    def per_frame(key, c="SQ_INSTS_VALU"):
        return rows[key][c] / FRAMES
    full = per_frame(("analyze", 0))
    print("k_analyze_v2<8,true,8>: wave-level instructions per frame (SQ counters per launch / %d frames), 8 ch x 60 s" % FRAMES)
    print("  whole kernel            VALU %7.1f   LDS %6.1f   SALU %6.1f   VMEM %5.1f" % (full, per_frame(("analyze", 0), "SQ_INSTS_LDS"),
          per_frame(("analyze", 0), "SQ_INSTS_SALU"), per_frame(("analyze", 0), "SQ_INSTS_VMEM")))
    for abl, what in ((1, "FFT passes (window, 3 passes, twiddles)"), (2, "atan2 + magnitude"), (4, "phase difference, wrap, frequency"),
                      (8, "MF stores (address arithmetic, packing)"), (16, "sample loads"), (32, "mirror read (LDS) of the pair split")):
        if ("analyze", abl) in rows:
            d = full - per_frame(("analyze", abl))
            print("  %-42s VALU %7.1f  (%4.1f %%)" % (what, d, 100.0 * d / full))
    if ("analyze", 63) in rows:
        d = per_frame(("analyze", 63))
        print("  %-42s VALU %7.1f  (%4.1f %%)" % ("left with all six off: split, sums, NaN max, loop", d, 100.0 * d / full))
    if ("synthesize", 0) in rows:
        fs = per_frame(("synthesize", 0))
        print("k_synthesize_v2<8,4>: whole kernel VALU %7.1f per frame   LDS %6.1f   SALU %6.1f   VMEM %5.1f" % (fs, per_frame(("synthesize", 0), "SQ_INSTS_LDS"),
              per_frame(("synthesize", 0), "SQ_INSTS_SALU"), per_frame(("synthesize", 0), "SQ_INSTS_VMEM")))
        if ("synthesize", 8) in rows:
            d = per_frame(("synthesize", 8))
            print("  %-42s VALU %7.1f  (%4.1f %%)" % ("bins: phase add, fold, sincos, pair merge", d, 100.0 * d / fs))
            print("  %-42s VALU %7.1f  (%4.1f %%)" % ("transform: 3 passes, window, overlap-add", fs - d, 100.0 * (fs - d) / fs))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--report")
    a = ap.parse_args()
    report(a.report) if a.report else run()
