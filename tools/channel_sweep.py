#!/usr/bin/env python3
"""Eight minutes of audio cut into 8 ... 512 channels: round trips at three sizes.  Channel counts that do not divide the kernels' wavefront slots (12, 24, 48, 96) used to
cost a second, nearly empty round of blocks (round 6: core.hip choose_chain_length counts BLOCKS now) -- profiles/r06_channel_sweep.txt."""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import flan_amd as fa
dev = torch.device("cuda", 0); SR = 48000.0
for (W, hop, dft) in ((2048, 512, 2048), (512, 128, 512), (2048, 512, 4096)):
    for ch in (8, 12, 16, 24, 32, 48, 63, 64, 96, 128, 512):
        n = int(8 * 60 * SR / ch)
        F = int(fa.lib.flanhip_num_pv_frames(n, hop)); bins = dft // 2 + 1
        audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
        fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 5, None))
        pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
        out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, SR / hop, W), dtype=torch.uint8, device=dev)
        def rt():
            fa.analyze_dev_fused(audio, ch, n, SR, W, hop, dft, pv, ws, None)
            fa.synthesize_dev_fused(pv, ch, F, bins, SR, SR / hop, W, out, ws, None, None)
        for _ in range(3): rt()
        torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 20
        for _ in range(reps): rt()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("(%d, %d, %d) %4d ch x %7.2f s  %8.4f ms  %7.1f M frames/s" % (W, hop, dft, ch, n / SR, ms, ch * F / ms / 1e3), flush=True)
        del audio, pv, out, ws
