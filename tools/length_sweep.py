#!/usr/bin/env python3
"""Round trips over input LENGTH at 1, 2 and 8 channels (three sizes): M frames/s should rise monotonically towards the plateau; a dip is an auxiliary kernel (scan,
sums, fix-up) or a chain cut gone wrong at that layout.  python tools/length_sweep.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa
dev = torch.device("cuda", 0); SR = 48000.0
for (W, hop, dft) in ((2048, 512, 2048), (512, 128, 512), (2048, 128, 4096), (4096, 1024, 8192)):
    for ch in (1, 2, 8):
        for seconds in (2, 5, 10, 20, 40, 80, 160, 320, 640, 1280):
            n = int(seconds * SR)
            F = int(fa.lib.flanhip_num_pv_frames(n, hop)); bins = dft // 2 + 1
            if ch * F * bins * 8 > 20e9:
                continue
            audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
            fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 5, None))
            pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
            out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
            ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, SR / hop, W), dtype=torch.uint8, device=dev)
            def rt():
                fa.analyze_dev_fused(audio, ch, n, SR, W, hop, dft, pv, ws, None)
                fa.synthesize_dev_fused(pv, ch, F, bins, SR, SR / hop, W, out, ws, None, None)
            for _ in range(3): rt()
            torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 0
            while reps < 5 or (time.perf_counter() - t0 < 0.03 and reps < 100):
                rt(); reps += 1
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            print("(%d, %d, %d) %2d ch x %5d s  %9d frames  %9.4f ms  %7.1f M frames/s" % (W, hop, dft, ch, seconds, ch * F, ms, ch * F / ms / 1e3), flush=True)
            del audio, pv, out, ws
