#!/usr/bin/env python3
"""Where does a batch of 1024 half-second channels lose its time?  The same blocks and chains in 1024, 128 and 8 channels, analysis and synthesis (the call, scan kernels
included) timed apart: the analysis does not care, the synthesis call did -- k_phase_scan2 in front of it took 0.39 ms for 1024 channels of 8 chains (round 6: the flat scan now)."""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import flan_amd as fa
dev = torch.device("cuda", 0)
SR = 48000.0
W, hop, dft = 2048, 512, 2048
def run(ch, n, chain_len, tag):
    F = int(fa.lib.flanhip_num_pv_frames(n, hop)); bins = dft // 2 + 1
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 5, None))
    pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
    with fa.debug_options(chain_len=chain_len):
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, SR / hop, W), dtype=torch.uint8, device=dev)
        def ana(): fa.analyze_dev_fused(audio, ch, n, SR, W, hop, dft, pv, ws, None)
        def syn(): fa.synthesize_dev_fused(pv, ch, F, bins, SR, SR / hop, W, out, ws, None, None)
        res = []
        for fn in (ana, syn):
            ana(); syn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ana()
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20)
    print("%-34s %5d ch x %6d frames/ch  chain_len %2d   analysis %.4f ms  synthesis %.4f ms" % (tag, ch, F, chain_len, res[0], res[1]), flush=True)
run(1024, 22560, 6, "1024 short channels")
run(128, 180480, 6, "128 channels, same blocks / chains")
run(8, 2880000, 6, "8 channels, same chain length")
run(8, 2880000, 0, "8 channels, library's cut")
run(1024, 22560, 0, "1024 short channels, library's cut")
