#!/usr/bin/env python3
"""Where the CPU checker's time goes on this host: analysis and synthesis per frame, and its FFT alone (one plan, r2c + c2r)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
lib = O._load()
lib.oracle_fft_pairs.restype = ctypes.c_double
lib.oracle_fft_pairs.argtypes = [ctypes.c_int, ctypes.c_int]
x = O.noise(1, 48000 * 20, seed=1)
for i in range(3):
    t0 = time.perf_counter(); pv = O.analyze(x, 48000.0, 2048, 512, 2048); t1 = time.perf_counter()
    y = O.synthesize(pv, 48000.0, 48000.0 / 512, 2048); t2 = time.perf_counter()
    lib.oracle_fft_pairs(2048, 2000); t3 = time.perf_counter()
    print("analysis %.1f us/frame  synthesis %.1f us/frame  fft pair %.1f us" % ((t1 - t0) / pv.shape[1] * 1e6, (t2 - t1) / pv.shape[1] * 1e6, (t3 - t2) / 2000 * 1e6))
