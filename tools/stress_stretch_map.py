#!/usr/bin/env python3
"""k_stretch_map (hand-counted waits, three tiles of rows in flight) against the oracle on a few hundred random grid shapes, one process:
    python tools/stress_stretch_map.py [iterations]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flan_amd as fa
import oracle_lib as O

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(7)
lib = fa.lib
bad = 0
for it in range(n_iter):
    F = int(rng.choice([1, 2, 3, 223, 224, 225, 447, 448, 449, 671, 672, 673])) if rng.random() < 0.3 else int(rng.integers(1, 3000))
    bins = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 1025, 2049])) if rng.random() < 0.4 else int(rng.integers(1, 1300))
    g = rng.uniform(0.05, 4.0, (F, bins)).astype(np.float32)
    ref = O.stretch_map(g, 48000.0, 256)
    d = ctypes.c_void_p(); fa.check(lib.flanhip_malloc(ctypes.byref(d), g.nbytes))
    dm = ctypes.c_void_p(); fa.check(lib.flanhip_malloc(ctypes.byref(dm), 4))
    fa.check(lib.flanhip_memcpy_h2d(d, g.ctypes.data_as(ctypes.c_void_p), g.nbytes, None))
    wide = int(rng.random() < 0.25)
    with fa.debug_options(wide_offsets=wide):
        fa.check(lib.flanhip_stretch_map_dev(d, F, bins, 48000.0, 256, dm, None))
    got = np.empty_like(g); mx = np.empty(1, np.float32)
    fa.check(lib.flanhip_memcpy_d2h(got.ctypes.data_as(ctypes.c_void_p), d, g.nbytes, None))
    fa.check(lib.flanhip_memcpy_d2h(mx.ctypes.data_as(ctypes.c_void_p), dm, 4, None))
    fa.check(lib.flanhip_stream_synchronize(None))
    lib.flanhip_free(d); lib.flanhip_free(dm)
    ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and mx[0] == ref.max()
    if not ok:
        bad += 1
        print("MISMATCH at F=%d bins=%d wide=%d: %d words differ" % (F, bins, wide, int(np.sum(got.view(np.uint32) != ref.view(np.uint32)))))
print("%d shapes, %d mismatches" % (n_iter, bad))
sys.exit(1 if bad else 0)
