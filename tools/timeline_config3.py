#!/usr/bin/env python3
"""BASELINE config 3 as bench.py runs it (8 ch x 60 s: convert_to_PV -> stretch x2 -> convert_to_audio), a few back-to-back iterations:
meant to run under `rocprofv3 --kernel-trace` (tools/scripts/timeline_config3.sh prints one iteration's dispatches with durations and gaps);
on its own it prints the event-timed ms per iteration."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa

SR, W, HOP, DFT = 48000.0, 2048, 512, 2048
BINS = DFT // 2 + 1
ch, n = 8, 60 * 48000
lib = fa.lib
if "--scan" in sys.argv:                       # A/B: the scan over all the chains in front of the synthesis instead of the group totals (syn_variant 2)
    lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, 2)
dev = torch.device("cuda", 0)
F = int(lib.flanhip_num_pv_frames(n, HOP))
Fo = 2 * F
ar = SR / HOP
P = lambda t: ctypes.c_void_p(t.data_ptr())
audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
fa.check(lib.flanhip_noise_dev(P(audio), ch, n, 1234, None))
pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
grid = torch.empty((F, BINS), dtype=torch.float32, device=dev)
dmax = torch.empty(1, dtype=torch.float32, device=dev)
st = torch.empty((ch, Fo, BINS, 2), dtype=torch.float32, device=dev)
out = torch.empty((ch, Fo * HOP), dtype=torch.float32, device=dev)
ws = torch.empty(fa.synthesize_workspace_bytes(ch, Fo, BINS, SR, ar, W), dtype=torch.uint8, device=dev)


def config3():
    fa.check(lib.flanhip_analyze_dev(P(audio), ch, n, SR, W, HOP, DFT, P(pv), None))
    if "--grid" in sys.argv:                   # the general path (a sampled callable): a filled grid, scanned
        fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None))
        fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, SR, HOP, P(dmax), None))
    else:                                      # bench.py's config 3: the constant factor's map in closed form
        fa.check(lib.flanhip_stretch_map_const_dev(2.0, P(grid), F, BINS, SR, HOP, P(dmax), None))
    fa.check(lib.flanhip_modify_time_dev_fused(P(pv), ch, F, BINS, SR, ar, P(grid), Fo, P(st), W, P(ws), None))
    fa.check(lib.flanhip_synthesize_dev_fused_checked(P(st), ch, Fo, BINS, SR, ar, W, P(out), P(ws), None, None))


t_end = time.perf_counter() + 0.1
while time.perf_counter() < t_end:
    config3(); torch.cuda.synchronize()
ms = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        config3()
    e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1) / 10)
print("config 3: ms per iteration", [round(m, 4) for m in ms])
