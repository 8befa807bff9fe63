#!/usr/bin/env python3
"""Where are the cliffs?  Round trips (fused analysis -> synthesis) over a grid of plausible shapes -- dft a power of two 256 ... 32768, window = dft, dft / 2, dft / 4,
hop = window / 2 ... window / 32 -- on one input (default 4 ch x 30 s), in one process: G bins/s per shape, so that a shape off every tuned grid shows as an outlier.

    python tools/shape_sweep.py [channels] [seconds] [odd: windows / hops off every grid, dft sizes that are no power of two]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
SR = 48000.0
dev = torch.device("cuda", 0)
n = int(seconds * SR)
audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 99, None))
rows = []
shapes = []
if len(sys.argv) > 3 and sys.argv[3] == "odd":
    # windows and hops off every grid, dft sizes that are no power of two (mixed radix, chirp-z, residue pairs, and one size only the direct sums serve)
    for dft in [int(v) for v in os.environ.get("SWEEP_DFTS", "1024,2048,4096,8192,16384,32768,3000,4410,6000,10000,12000,15000,20000,22050,44100,48000,2998,8186,9998").split(",")]:
        for W in sorted({min(dft, 1000), min(dft, 2000), (dft // 2) | 1 if dft > 8 else dft, dft - 2 if dft % 4 else dft - 6, dft}):
            for hop in (W // 4, max(W // 7, 1), 441 if W >= 882 else W // 3):
                shapes.append((dft, W, hop))
else:
    for lg in range(int(os.environ.get("SWEEP_LG0", "8")), int(os.environ.get("SWEEP_LG1", "16"))):
        for wd in (1, 2, 4):
            for hd in (2, 4, 8, 16, 32):
                shapes.append((1 << lg, (1 << lg) // wd, ((1 << lg) // wd) // hd))
for (dft, W, hop) in shapes:
    if hop < 2:
        continue
    F = int(fa.lib.flanhip_num_pv_frames(n, hop))
    bins = dft // 2 + 1
    if ch * F * bins * 8 > 12e9:
        continue
    ar = SR / hop
    pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
    wsb = fa.synthesize_workspace_bytes(ch, F, bins, SR, ar, W)
    if wsb == 0:                                           # refused (dft x window overflows the int product of AudioPV.cpp:99 ...)
        print("dft %5d  W %5d  hop %5d   refused: %s" % (dft, W, hop, fa.last_error()), flush=True)
        continue
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def rt():
        fa.analyze_dev_fused(audio, ch, n, SR, W, hop, dft, pv, ws, None)
        fa.synthesize_dev_fused(pv, ch, F, bins, SR, ar, W, out, ws, None, None)
    rt()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < 0.05 and reps < 50):
        rt()
        reps += 1
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    gb = ch * F * bins / ms / 1e6
    rows.append((dft, W, hop, ms, gb))
    print("dft %5d  W %5d  hop %5d   %9.3f ms  %7.1f G bins/s%s" % (dft, W, hop, ms, gb, "   <-- " if gb < 60 else ""), flush=True)
    del pv, out, ws
