"""The checker's FFT (oracle/flan_oracle.cpp: FFTPlan, r2c, c2r) against numpy.fft in float64, for the sizes the reference can be called
with: FFTHelper.cpp:16-26 hands ANY dft_size to FFTW, so besides the radix-2 path (powers of two) the checker carries a mixed-radix path
over the prime factors.  Results leave the checker rounded once to fp32, like FFTW3f's: the bound below is that rounding."""
import numpy as np
import pytest

import oracle_lib as O

SIZES = [4, 6, 10, 30, 48, 100, 1000, 1024, 2998, 3000, 4094, 4096, 6000, 16384]      # 2998 = 2 x 1499 (a large prime factor)


@pytest.mark.parametrize("n", SIZES)
def test_r2c_and_c2r_against_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    X = np.zeros((n // 2 + 1) * 2, np.float32)
    assert O.lib.oracle_r2c(x, n, X) == 0
    got = X[0::2].astype(np.float64) + 1j * X[1::2].astype(np.float64)
    ref = np.fft.rfft(x.astype(np.float64))
    assert np.abs(got - ref).max() <= 1.3e-7 * np.abs(ref).max()                      # one fp32 rounding of an fp64 result
    assert X[1] == 0.0 and X[-1] == 0.0                                                # r2c: X[0] and X[N/2] are real (exact zeros)
    y = np.zeros(n, np.float32)
    assert O.lib.oracle_c2r(X, n, y) == 0
    back = np.fft.irfft(got, n) * n                                                    # c2r is unnormalised
    assert np.abs(y - back).max() <= 1.3e-7 * np.abs(back).max()


def test_c2r_ignores_the_imaginary_parts_of_dc_and_nyquist():
    n = 3000
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n // 2 + 1) * 2).astype(np.float32)
    Y = X.copy()
    Y[1] = 123.0
    Y[-1] = -7.0
    a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
    O.lib.oracle_c2r(X, n, a)
    O.lib.oracle_c2r(Y, n, b)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("W,hop,dft", [(2048, 512, 3000), (600, 150, 1000), (4096, 1024, 16384), (64, 16, 66)])
def test_analysis_of_a_sine_at_any_dft_size(W, hop, dft):
    """SURVEY 8c style anchor at a size without a radix-2 transform: a stationary sine comes out of convert_to_PV with its own frequency
    in the bins around it (steady-state frames), whatever the dft size"""
    sr = 48000.0
    freq = (dft // 8 + 0.5) * sr / dft                                                 # between two bins, well inside the spectrum
    n = 40 * hop + W
    x = (0.5 * np.sin(2 * np.pi * freq * np.arange(n) / sr)).astype(np.float32)[None, :]
    pv = O.analyze(x, sr, W, hop, dft)
    assert pv.shape == (1, n // hop + 1, dft // 2 + 1, 2)
    k = int(round(freq * dft / sr))
    mid = pv[0, 10:25, k, :]
    assert np.abs(mid[:, 1] - freq).max() <= 0.02 * sr / dft                           # f of the loudest bin: the sine's frequency
    out, flag = O.synthesize(pv, sr, np.float32(sr) / np.float32(hop), W)
    assert flag == 0 and out.shape == (1, (n // hop + 1) * hop)
