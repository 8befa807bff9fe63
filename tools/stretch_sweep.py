#!/usr/bin/env python3
"""BASELINE config 3's pipeline (convert_to_PV -> stretch -> convert_to_audio) at other sizes, factors and input shapes: ms per stage and G output MFs/s of the
modify_time stage.  A size or a factor whose stretch stage falls far below the dft 2048 x 2 figure would be a path off the tuned one.   python tools/stretch_sweep.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa
lib = fa.lib
dev = torch.device("cuda", 0); SR = 48000.0
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for (W, hop, dft) in ((2048, 512, 2048), (2048, 128, 4096), (512, 128, 512), (4096, 1024, 8192), (2048, 512, 3000)):
    for (ch, seconds, factor) in ((8, 60.0, 2.0), (8, 60.0, 0.5), (8, 60.0, 1.37), (2, 10.0, 2.0), (1, 3.0, 4.0)):
        n = int(seconds * SR)
        F = int(lib.flanhip_num_pv_frames(n, hop)); bins = dft // 2 + 1; ar = SR / hop
        if ch * F * bins * 8 * (1 + factor) > 30e9:
            continue
        audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
        fa.check(lib.flanhip_noise_dev(P(audio), ch, n, 5, None))
        pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
        grid = torch.empty((F, bins), dtype=torch.float32, device=dev)
        dmax = torch.empty(1, dtype=torch.float32, device=dev)
        fa.check(lib.flanhip_analyze_dev(P(audio), ch, n, SR, W, hop, dft, P(pv), None))
        fa.check(lib.flanhip_stretch_map_const_dev(factor, P(grid), F, bins, SR, hop, P(dmax), None))
        torch.cuda.synchronize()
        Fo = max(int(factor * F), 4)                                # (what PV::stretch asks for: PVModify.cpp:371-385; bench.py: 2 F at factor 2)
        st = torch.empty((ch, Fo, bins, 2), dtype=torch.float32, device=dev)
        out = torch.empty((ch, Fo * hop), dtype=torch.float32, device=dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, Fo, bins, SR, ar, W), dtype=torch.uint8, device=dev)
        t_map = timed(lambda: fa.check(lib.flanhip_stretch_map_const_dev(factor, P(grid), F, bins, SR, hop, P(dmax), None)))
        t_mod = timed(lambda: fa.check(lib.flanhip_modify_time_dev_fused(P(pv), ch, F, bins, SR, ar, P(grid), Fo, P(st), W, P(ws), None)))
        def syn():
            fa.check(lib.flanhip_modify_time_dev_fused(P(pv), ch, F, bins, SR, ar, P(grid), Fo, P(st), W, P(ws), None))
            fa.check(lib.flanhip_synthesize_dev_fused_checked(P(st), ch, Fo, bins, SR, ar, W, P(out), P(ws), None, None))
        t_both = timed(syn)
        print("(%d, %d, %d) %d ch x %5.1f s  x%.2f  F %6d -> %6d   map %.4f  modify_time %.4f ms (%.1f G MFs/s out)  + synthesis %.4f ms" % (W, hop, dft, ch, seconds, factor, F, Fo, t_map, t_mod, ch * Fo * bins / t_mod / 1e6, t_both - t_mod), flush=True)
        del audio, pv, grid, st, out, ws
