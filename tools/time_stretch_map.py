#!/usr/bin/env python3
"""flanhip_stretch_map_dev alone (config 3's map: 5626 frames x 1025 bins), events on the null stream: with and without the maximum, behind a
fill of the grid (as in the bench) or behind an idle stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flan_amd as fa
lib = fa.lib
dev = torch.device("cuda", 0)
F, BINS, SR, HOP = 5626, 1025, 48000.0, 512
grid = torch.empty((F, BINS), dtype=torch.float32, device=dev)
dmax = torch.empty(1, dtype=torch.float32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run(with_max, with_fill, reps=20):
    ms = []
    for _ in range(reps):
        if with_fill:
            fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None))
        else:
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, SR, HOP, P(dmax) if with_max else None, None))
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2]
for _ in range(50):
    fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, SR, HOP, P(dmax), None))
torch.cuda.synchronize()
for with_fill in (False, True):
    for with_max in (False, True):
        print("behind %s, %s the maximum: %.1f us" % ("a fill of the grid" if with_fill else "an idle stream", "with" if with_max else "without", 1e3 * run(with_max, with_fill)))
fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None))
fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, SR, HOP, P(dmax), None))
torch.cuda.synchronize()
print("maximum:", float(dmax.item()), "expected", float(grid.max().item()))
