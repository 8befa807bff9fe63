#!/usr/bin/env python3
"""The headline round trip at sample rates 8 ... 192 kHz (and two odd ones) x hops 512 / 441 / 128: the analysis rate sr / hop is a divisor the kernels divide by
through a proven reciprocal pair or not at all (pv_math.h: div_c); a rate without an exact plan must not be a cliff.  (Round 6: 163-181 M frames/s everywhere.)"""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import flan_amd as fa
dev = torch.device("cuda", 0)
W, hop, dft, ch, n = 2048, 512, 2048, 8, 2880000
F = int(fa.lib.flanhip_num_pv_frames(n, hop)); bins = dft // 2 + 1
audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 5, None))
pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
for SR in (8000.0, 11025.0, 16000.0, 22050.0, 32000.0, 44100.0, 48000.0, 88200.0, 96000.0, 192000.0, 12345.0, 47999.0):
    for hp in (512, 441, 128):
        Fh = int(fa.lib.flanhip_num_pv_frames(n, hp))
        pvh = torch.empty((ch, Fh, bins, 2), dtype=torch.float32, device=dev)
        outh = torch.empty((ch, Fh * hp), dtype=torch.float32, device=dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, Fh, bins, SR, SR / hp, W), dtype=torch.uint8, device=dev)
        def rt():
            fa.analyze_dev_fused(audio, ch, n, SR, W, hp, dft, pvh, ws, None)
            fa.synthesize_dev_fused(pvh, ch, Fh, bins, SR, SR / hp, W, outh, ws, None, None)
        for _ in range(3): rt()
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
        for _ in range(reps): rt()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("sr %8g hop %4d  %8.4f ms  %7.1f M frames/s" % (SR, hp, ms, ch * Fh / ms / 1e3), flush=True)
        del pvh, outh, ws
