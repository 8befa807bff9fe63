#!/usr/bin/env python3
"""A/B in one process, interleaved rounds: the overlaps of neighbouring chains added inside k_synthesize_v2 (the inline_fixup hook) against k_ola_fixup as a
launch of its own (the default), fused round trip at several shapes.  ms per step, median and minimum."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa

SR, W, HOP, DFT = 48000.0, 2048, 512, 2048
if len(sys.argv) > 1:                          # another shape: tools/ab_fixup.py WINDOW HOP DFT
    W, HOP, DFT = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
BINS = DFT // 2 + 1
dev = torch.device("cuda", 0)
res = {}
for ch, seconds in ((1, 5), (8, 60), (2, 60), (8, 600)):
    n = int(seconds * SR)
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    ar = SR / HOP
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, None))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)

    def step():
        fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, None)
        fa.synthesize_dev_fused(pv, ch, F, BINS, SR, ar, W, out, ws, None, None)
    for _ in range(200 if seconds < 100 else 20):
        step()
    torch.cuda.synchronize()
    t = {0: [], 1: []}
    reps = 20 if seconds < 100 else 5
    for r in range(9):
        for mode in (0, 1):
            fa.lib.flanhip_debug_option(fa.DEBUG_INLINE_FIXUP, 1 + mode)        # 1: inside the kernel, 2: never
            step(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                step()
            e1.record(); torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / reps)
    fa.lib.flanhip_debug_option(fa.DEBUG_INLINE_FIXUP, 0)
    res["%d ch x %d s" % (ch, seconds)] = {("inside the kernel" if m == 0 else "separate launch"): {"median_ms": round(sorted(v)[len(v) // 2], 4), "min_ms": round(min(v), 4)} for m, v in t.items()}
    del audio, pv, out, ws
print(json.dumps(res, indent=1))
