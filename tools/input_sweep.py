#!/usr/bin/env python3
"""The headline round trip (2048, 512, 2048) -- and the API default (2048, 128, 4096) -- over inputs of different SHAPE: one long channel, many short ones, tiny ones.
M frames/s per input; what is far below the 8 ch x 60 s figure at a comparable frame count would be a partitioning problem (choose_chain_length).

    python tools/input_sweep.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa

SR = 48000.0
dev = torch.device("cuda", 0)
for (W, hop, dft) in ((2048, 512, 2048), (2048, 128, 4096), (512, 128, 512), (8192, 2048, 8192)):
    for (ch, seconds) in ((8, 60.0), (1, 480.0), (1, 60.0), (64, 7.5), (256, 1.875), (1024, 0.47), (2, 5.0), (1, 1.0), (1, 0.1), (3, 33.3), (7, 11.0)):
        n = int(seconds * SR)
        F = int(fa.lib.flanhip_num_pv_frames(n, hop))
        bins = dft // 2 + 1
        audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
        fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 5, None))
        pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
        out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, SR / hop, W), dtype=torch.uint8, device=dev)

        def rt():
            fa.analyze_dev_fused(audio, ch, n, SR, W, hop, dft, pv, ws, None)
            fa.synthesize_dev_fused(pv, ch, F, bins, SR, SR / hop, W, out, ws, None, None)
        for _ in range(3):
            rt()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 0
        while reps < 5 or (time.perf_counter() - t0 < 0.05 and reps < 200):
            rt()
            reps += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("(%d, %d, %d)  %4d ch x %7.2f s  %8d frames  %8.4f ms  %7.1f M frames/s" % (W, hop, dft, ch, seconds, ch * F, ms, ch * F / ms / 1e3), flush=True)
        del audio, pv, out, ws
