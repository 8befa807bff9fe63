#!/usr/bin/env python3
"""flanhip_stretch_map_const_dev against its yardsticks (flanhip_fill_dev of the same grid; fill + the scanning kernel), event-timed, back to back."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa
lib = fa.lib
F, BINS = 5626, 1025
dev = torch.device("cuda", 0)
grid = torch.empty((F, BINS), dtype=torch.float32, device=dev)
dmax = torch.empty(1, dtype=torch.float32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("fill            %.2f us" % timeit(lambda: fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None))))
print("const map       %.2f us" % timeit(lambda: fa.check(lib.flanhip_stretch_map_const_dev(2.0, P(grid), F, BINS, 48000.0, 512, P(dmax), None))))
print("const map 1.3   %.2f us" % timeit(lambda: fa.check(lib.flanhip_stretch_map_const_dev(1.3, P(grid), F, BINS, 48000.0, 512, P(dmax), None))))
def scan():
    fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None)); fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, 48000.0, 512, P(dmax), None))
print("fill + scan     %.2f us" % timeit(scan))
