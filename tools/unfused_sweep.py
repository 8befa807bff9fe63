#!/usr/bin/env python3
"""The fused round trip (the analysis kernel leaves the synthesis its chain sums) against the unfused pair of calls (k_phase_sums2 reads the PV again) over sizes and
input shapes: unfused 1.04-1.39 x everywhere but for ONE short clip at the API default ((2048, 128, 4096), 2 ch x 5 s: fused 0.196 ms, unfused 0.150 -- the fused analysis
walks the synthesis' chains of at least window / hop - 1 = 15 frames).  Round 6."""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import flan_amd as fa
dev = torch.device("cuda", 0); SR = 48000.0
for (W, hop, dft) in ((2048, 512, 2048), (512, 128, 512), (2048, 128, 4096), (4096, 1024, 8192), (2048, 512, 3000)):
    for (ch, seconds) in ((8, 60.0), (1, 480.0), (64, 7.5), (1024, 0.47), (2, 60.0), (2, 5.0)):
        n = int(seconds * SR)
        F = int(fa.lib.flanhip_num_pv_frames(n, hop)); bins = dft // 2 + 1
        audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
        fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 5, None))
        pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
        out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, SR / hop, W), dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        def fused():
            fa.analyze_dev_fused(audio, ch, n, SR, W, hop, dft, pv, ws, None)
            fa.synthesize_dev_fused(pv, ch, F, bins, SR, SR / hop, W, out, ws, None, None)
        def unfused():
            fa.analyze_dev(audio, ch, n, SR, W, hop, dft, pv)
            fa.synthesize_dev(pv, ch, F, bins, SR, SR / hop, W, out, ws, flag)
        res = []
        for fn in (fused, unfused):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / reps * 1e3)
        print("(%d, %d, %d) %4d ch x %7.2f s  fused %8.4f ms  unfused %8.4f ms  x%.2f" % (W, hop, dft, ch, seconds, res[0], res[1], res[1] / res[0]), flush=True)
        del audio, pv, out, ws
