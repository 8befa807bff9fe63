#!/usr/bin/env python3
"""The 2:1 block convolver's two generations in one process, interleaved rounds: k_resample_ols3 (512 threads, radix 8: the default) and
k_resample_ols2 (256 threads, radix 16: the resample_direct = 2 hook), 2 ch x 60 s at 96 kHz -> 48 kHz; and how their outputs differ."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import flan_amd as fa

fa.check(fa.lib.flanhip_set_device(0))
dev = torch.device("cuda", 0)
src, dst, ch = 96000.0, 48000.0, 2
n = int(60 * src)
n_out = int(fa.lib.flanhip_resample_out_frames(n, src, dst))
x = torch.empty((ch, n), dtype=torch.float32, device=dev)
fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(x.data_ptr()), ch, n, 7, None))
y = torch.empty((ch, n_out), dtype=torch.float32, device=dev)
call = lambda: fa.check(fa.lib.flanhip_resample_dev(ctypes.c_void_p(x.data_ptr()), ch, n, src, dst, ctypes.c_void_p(y.data_ptr()), None))
outs, ms = {}, {0: [], 2: []}
for v in (0, 2):
    fa.lib.flanhip_debug_option(fa.DEBUG_RESAMPLE_DIRECT, v)
    call(); torch.cuda.synchronize()
    outs[v] = y.clone()
for _ in range(200):
    call()
for r in range(9):
    for v in (0, 2):
        fa.lib.flanhip_debug_option(fa.DEBUG_RESAMPLE_DIRECT, v)
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        ms[v].append(e0.elapsed_time(e1) / 20)
fa.lib.flanhip_debug_option(fa.DEBUG_RESAMPLE_DIRECT, 0)
d = (outs[0].double() - outs[2].double()).abs()
same = float((outs[0].view(torch.int32) == outs[2].view(torch.int32)).double().mean().item())
for v, name in ((0, "k_resample_ols3"), (2, "k_resample_ols2")):
    m = sorted(ms[v])
    print("%s: median %.4f ms  min %.4f ms" % (name, m[len(m) // 2], m[0]))
print("outputs: bit-identical share %.6f, max abs difference %.3e" % (same, float(d.max().item())))
