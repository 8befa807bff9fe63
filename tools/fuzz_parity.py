#!/usr/bin/env python3
"""A one-off randomised parity run, wider than the suite's sweeps: N random calls (channels, length, window, hop, dft size drawn so that EVERY kernel family is hit --
the tuned sizes on and off their grids, fractions of a step at dft 8192 / 16384, mixed-radix, chirp-z in LDS and in device memory, residue pairs plain and mixed, direct
sums; 1 ... 40 channels, a few frames ... a few hundred) through the C ABI against the CPU oracle (tests/oracle_lib.py: test infrastructure, used here as the checker):
P1 rel_m <= 1e-5, weighted f error <= 2e-3 Hz, P2 (the oracle's PV into both synthesisers) RMS <= 1e-5.   python tools/fuzz_parity.py [N=150] [seed=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import flan_amd as fa
from test_gpu_conversions import p1_metrics

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sr = 48000.0
pow2 = [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
other = [3000, 1000, 4410, 6000, 12000, 15000, 2998, 5998, 8186, 9998, 10002, 20000, 22050, 24576, 30002, 44100, 66, 134]
worst = {"rel_m": 0.0, "wrms": 0.0, "rms": 0.0, "same": 1.0}
t_start = time.time()
for it in range(N):
    dft = int(rng.choice(pow2)) if rng.random() < 0.65 else int(rng.choice(other))
    r = rng.random()
    if r < 0.5:                                                    # on the tuned grids: window = dft / 2^a, hop = window / 2^b
        W = max(dft >> int(rng.integers(0, 3)), 4) if dft & (dft - 1) == 0 else min(dft, 1 << int(rng.integers(6, 12)))
        hop = max(W >> int(rng.integers(1, 6)), 1)
    else:                                                          # anything
        W = int(rng.integers(4, dft + 1))
        hop = int(rng.integers(1, W + 1))
    if dft * W >= 2 ** 31:
        continue
    ch = int(rng.choice([1, 1, 2, 3, 8, 17, 40]))
    frames = int(rng.integers(1, 260))
    n = max(hop * frames + int(rng.integers(0, hop)), 3)
    if ch * (n // hop + 1) * (dft // 2 + 1) > 6e7 or ch * n > 4e7:
        ch = 1
        if (n // hop + 1) * (dft // 2 + 1) > 6e7:
            continue
    c = dft // 2
    for q in (2, 3, 5, 7, 11, 13):
        while c % q == 0:
            c //= q
    if c > 1 and ch * (n // hop + 1) * float(dft // 2) ** 2 > 2e10:     # (the oracle's transform of a size with a large prime factor is O( N^2 ): minutes on one core)
        ch = 1
        if (n // hop + 1) * float(dft // 2) ** 2 > 2e10:
            continue
    x = O.noise(ch, n, seed=1000 + it)
    ar = np.float32(sr) / np.float32(hop)
    ref = O.analyze(x, sr, W, hop, dft)
    got = fa.analyze(x, sr, W, hop, dft)
    rel_m, wrms, same, turns = p1_metrics(got, ref, sr / hop)
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    out, _ = fa.synthesize(ref, sr, ar, W)
    rms = float(np.sqrt(np.mean((out.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
    worst["rel_m"] = max(worst["rel_m"], rel_m); worst["wrms"] = max(worst["wrms"], wrms); worst["rms"] = max(worst["rms"], rms); worst["same"] = min(worst["same"], same)
    ok = rel_m <= 1e-5 and wrms <= max(2e-3, 1e-7 * sr / hop * 20) and rms <= 1e-5 and np.all(np.isfinite(got)) and got.shape == ref.shape
    print("%3d  %2d ch  n %8d  (%5d, %5d, %6d)  rel_m %.2e  wrms_df %.2e  same %.4f  P2 rms %.2e  %s" % (it, ch, n, W, hop, dft, rel_m, wrms, same, rms, "" if ok else "<-- FAIL"), flush=True)
    if not ok:
        sys.exit(1)
print("worst:", worst, " %.0f s" % (time.time() - t_start))
