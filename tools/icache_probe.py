#!/usr/bin/env python3
"""Does a kernel launch start cold?  Time the analysis kernel (a) alone, back to back, (b) alternating with synthesis, per launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa
SR, W, HOP, DFT = 48000.0, 2048, 512, 2048
BINS = DFT // 2 + 1
ch, n = 8, 60 * 48000
dev = torch.device("cuda", 0)
F = int(fa.lib.flanhip_num_pv_frames(n, HOP)); ar = SR / HOP
audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 1234, None))
pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
A = lambda: fa.analyze_dev(audio, ch, n, SR, W, HOP, DFT, pv)
S = lambda: fa.synthesize_dev(pv, ch, F, BINS, SR, ar, W, out, ws, None)
def timed(seq, reps=30):
    for f in seq: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in seq: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for i in range(4):
    print("analysis alone, pass %d: %.4f ms" % (i, timed([A])))
tA, tS, tAS = timed([A]), timed([S]), timed([A, S])
print("analysis alone %.4f ms   synthesis alone %.4f ms   sum %.4f   alternating %.4f ms" % (tA, tS, tA + tS, tAS))
tAA = timed([A, A])
print("A,A pairs %.4f ms (per launch %.4f)" % (tAA, tAA / 2))
print("analysis alone again: %.4f ms;  reps=300: %.4f ms" % (timed([A]), timed([A], 300)))
print("alternating again: %.4f ms;  reps=300: %.4f ms" % (timed([A, S]), timed([A, S], 300)))
