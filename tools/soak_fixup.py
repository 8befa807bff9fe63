#!/usr/bin/env python3
"""Soak of the synchronisation inside the synthesis kernels (the chains' overlaps added by whichever chain finishes its half last: pv_kernels_v2.h /
_v3.h, round 6: pv_kernels_eo.h / _team.h): the same fused round trip launched N times per shape -- back to back, so that launches overlap on the device, and on two streams at once with
a workspace each -- every output compared BIT FOR BIT with the first launch's and with the separate-launch form (k_ola_fixup).  Any ordering the
protocol does not cover shows as a differing sample.

    python tools/soak_fixup.py [launches per shape, default 1500] [team: the round-6 shapes only]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
SR = 48000.0
dev = torch.device("cuda", 0)
shapes = [(8, 60.0, 2048, 512, 2048), (2, 60.0, 2048, 512, 2048), (3, 7.3, 2048, 512, 2048), (8, 600.0, 2048, 512, 2048), (1, 200.0, 2048, 128, 2048),
          (8, 60.0, 1024, 256, 1024), (5, 11.1, 1024, 512, 1024), (8, 60.0, 512, 128, 512), (2, 33.3, 512, 256, 512), (4, 20.0, 2048, 1024, 2048),
          # round 6: the team kernels' form of the protocol (a word per wavefront of a chain: pv_kernels_eo.h ChainOverlap) -- dft 4096 (hop 128 by half steps, 512,
          # window = dft), 8192, 16384
          (2, 60.0, 2048, 128, 4096), (8, 60.0, 2048, 512, 4096), (3, 21.0, 4096, 1024, 4096), (8, 60.0, 8192, 2048, 8192), (2, 60.0, 4096, 512, 8192),
          (8, 60.0, 4096, 1024, 16384), (2, 60.0, 16384, 4096, 16384)]
if len(sys.argv) > 2 and sys.argv[2] == "team":
    shapes = shapes[10:]
report = {}
for (ch, seconds, W, HOP, DFT) in shapes:
    n = int(seconds * SR)
    bins = DFT // 2 + 1
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    ar = SR / HOP
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 4321, None))
    launches = N if seconds < 100 else max(N // 20, 20)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bufs = []
    for s in streams:
        bufs.append((torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev), torch.empty((ch, F * HOP), dtype=torch.float32, device=dev),
                     torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, ar, W), dtype=torch.uint8, device=dev)))

    def step(i):
        pv, out, ws = bufs[i]
        st = int(streams[i].cuda_stream)
        fa.analyze_dev_fused(audio, ch, n, SR, W, HOP, DFT, pv, ws, st)
        fa.synthesize_dev_fused(pv, ch, F, bins, SR, ar, W, out, ws, None, st)
    # the separate-launch form as the yardstick
    fa.lib.flanhip_debug_option(fa.DEBUG_INLINE_FIXUP, 2)
    step(0); torch.cuda.synchronize()
    want = bufs[0][1].clone()
    fa.lib.flanhip_debug_option(fa.DEBUG_INLINE_FIXUP, 1)                     # inside the kernel, whatever the chain length
    bad = 0
    checked = 0
    for r in range(launches):
        step(0); step(1)
        if r % 8 == 7 or r == launches - 1:
            # (launches in between overwrite the same buffers while earlier ones may still run: what is compared is the last pair's output)
            torch.cuda.synchronize()
            for i in range(2):
                checked += 1
                if not torch.equal(bufs[i][1].view(torch.int32), want.view(torch.int32)):
                    bad += 1
    fa.lib.flanhip_debug_option(fa.DEBUG_INLINE_FIXUP, 0)
    report["%d ch x %g s (%d, %d, %d)" % (ch, seconds, W, HOP, DFT)] = {"launches": 2 * launches, "outputs compared": checked, "differing": bad}
    print(list(report.items())[-1], flush=True)
    del audio, bufs, want
print(json.dumps(report, indent=1))
sys.exit(1 if any(v["differing"] for v in report.values()) else 0)
