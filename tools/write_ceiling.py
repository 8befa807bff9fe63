#!/usr/bin/env python3
"""What a WRITE-ONLY stream reaches on this box (flanhip_fill_dev: 16 bytes per lane), beside the copy (flanhip_copy_dev: the same bytes read + written)
and a read-mostly reduction (flanhip_sqdiff_dev): the analysis kernels write 8200 of the 10248 bytes they move per frame."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa
dev = torch.device("cuda", 0)
n = 45008 * 1025 * 2                     # floats of the headline's PV (369 MB)
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
r = torch.zeros(2, dtype=torch.float64, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timed(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
t_fill = timed(lambda: fa.check(fa.lib.flanhip_fill_dev(P(a), n, 1.0, None)))
t_copy = timed(lambda: fa.check(fa.lib.flanhip_copy_dev(P(a), P(b), n, None)))
t_read = timed(lambda: fa.check(fa.lib.flanhip_sqdiff_dev(P(a), P(b), n, P(r), None)))
print("write only  %d MB: %.4f ms = %.2f TB/s" % (n * 4 // 1000000, t_fill, n * 4 / t_fill / 1e9))
print("copy        %d MB read + %d MB written: %.4f ms = %.2f TB/s" % (n * 4 // 1000000, n * 4 // 1000000, t_copy, 2 * n * 4 / t_copy / 1e9))
print("read only   2 x %d MB: %.4f ms = %.2f TB/s" % (n * 4 // 1000000, t_read, 2 * n * 4 / t_read / 1e9))
