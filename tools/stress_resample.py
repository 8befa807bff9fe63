#!/usr/bin/env python3
"""The 2:1 FFT convolver (k_resample_ols3) against the direct fp64 sums (resample_direct hook) on random stream lengths and channel counts, one process:
    python tools/stress_resample.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flan_amd as fa
import oracle_lib as O

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(11)
bad = 0
worst_share, worst_abs = 1.0, 0.0
for it in range(n_iter):
    ch = int(rng.integers(1, 4))
    n = int(rng.choice([19808, 19809, 24760, 24761, 39616, 4952 * 9, 4952 * 9 + 1])) if rng.random() < 0.3 else int(rng.integers(19808, 400000))
    x = O.noise(ch, n, seed=it + 100)
    with fa.debug_options(resample_direct=1):
        direct = fa.resample(x, 96000.0, 48000.0)
    got = fa.resample(x, 96000.0, 48000.0)
    same = float(np.mean(got.view(np.uint32) == direct.view(np.uint32)))
    worst = float(np.abs(got.astype(np.float64) - direct.astype(np.float64)).max())
    worst_share, worst_abs = min(worst_share, same), max(worst_abs, worst)
    if got.shape != direct.shape or same < 0.999 or worst > 1.2e-7:
        bad += 1
        print("MISMATCH at ch=%d n=%d: same %.5f worst %.2e" % (ch, n, same, worst))
print("%d streams, %d out of tolerance; lowest bit-identical share %.5f, largest difference %.2e" % (n_iter, bad, worst_share, worst_abs))
sys.exit(1 if bad else 0)
