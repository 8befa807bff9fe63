"""The CPU restatement (oracle/flan_oracle.cpp) against the REAL reference translation units that build here
unmodified (oracle/_ref/libflanref.so = phase_vocoder.cpp, WindowFunctions.cpp, PV/PVBuffer.cpp ... read in place
from /root/reference).  Bit-exact, millions of cases.  Skipped where oracle/_ref was never built."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

ref = O.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libflanref.so not built (no /root/reference)")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_pi2_constant():
    assert np.float32(ref.ref_pi2()).view(np.uint32) == np.float32(O.lib.oracle_pi2()).view(np.uint32)
    assert float(np.float32(ref.ref_pi2())) == 6.2831854820251465  # float(2*float(pi)), NOT 2*pi


def test_hann_bit_exact():
    # every window size the path uses + a dense sweep of [0,1]
    for W in (64, 256, 1024, 2048, 4096, 8192, 1000):
        ours = O.hann_window(W)
        theirs = np.array([ref.ref_hann(np.float32(i) / np.float32(W - 1)) for i in range(W)], np.float32)
        assert np.array_equal(_bits(ours), _bits(theirs)), W
    xs = np.random.default_rng(1).random(20000).astype(np.float32)
    a = np.array([O.lib.oracle_hann(float(x)) for x in xs], np.float32)
    b = np.array([ref.ref_hann(float(x)) for x in xs], np.float32)
    assert np.array_equal(_bits(a), _bits(b))


@pytest.mark.parametrize("hop,dft", [(512, 2048), (128, 4096), (2048, 2048), (1, 2048)])
def test_phase_vocoder_bit_exact(hop, dft):
    """phase_vocoder.cpp:5-53 over random spectra, random previous phases, every bin frequency."""
    rng = np.random.default_rng(hop * 7 + dft)
    n = 2_000_000
    sr = np.float32(48000.0)
    ar = np.float32(sr / np.float32(hop))
    re = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 3, n)).astype(np.float32)
    im = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 3, n)).astype(np.float32)
    # exact zeros, axis-aligned and denormal inputs
    re[:1000] = 0; im[:500] = 0; im[1000:1500] = 0; re[2000:2100] = 1e-42; im[2100:2200] = -1e-42
    bins = rng.integers(0, dft // 2 + 1, n)
    binf = np.array([0], np.float32)  # placeholder for dtype
    binf = (bins.astype(np.float32) * sr / np.float32(dft)).astype(np.float32)
    prev = np.float32(rng.uniform(-np.pi, np.pi, n)).astype(np.float64)
    prev[:100000] = 0.0
    pa, pb = prev.copy(), prev.copy()
    m_ref = np.empty(n, np.float32); f_ref = np.empty(n, np.float32)
    ref.ref_phase_vocoder_batch(n, pb, re, im, binf, ar, sr, m_ref, f_ref)
    m_o = np.empty(n, np.float32); f_o = np.empty(n, np.float32)
    mm = C.c_float(); ff = C.c_float(); ph = C.c_double()
    # batch through the scalar oracle entry point (vectorised by chunks to keep python overhead sane)
    step = 1
    idx = np.arange(0, n, 37)[:60000]  # 60k scalar probes ...
    for i in idx:
        ph.value = pa[i]
        O.lib.oracle_phase_vocoder(C.byref(ph), float(re[i]), float(im[i]), float(binf[i]), float(ar), float(sr), C.byref(mm), C.byref(ff))
        m_o[i] = mm.value; f_o[i] = ff.value; pa[i] = ph.value
    assert np.array_equal(_bits(m_o[idx]), _bits(m_ref[idx]))
    assert np.array_equal(_bits(f_o[idx]), _bits(f_ref[idx]))
    assert np.array_equal(pa[idx], pb[idx])


def test_phase_vocoder_through_frame_loop():
    """... and the full sweep: run the oracle's analysis loop and replay its (phase state, spectrum) stream through
    the real flan::phase_vocoder -- every bin of every frame must agree bit for bit."""
    x = O.noise(1, 20000, seed=5)
    sr, W, hop, dft = 48000.0, 2048, 512, 2048
    pv = O.analyze(x, sr, W, hop, dft)
    F, bins = pv.shape[1], pv.shape[2]
    w = O.hann_window(W)
    state = np.zeros(bins, np.float64)
    binf = (np.arange(bins, dtype=np.float32) * np.float32(sr) / np.float32(dft)).astype(np.float32)
    for fr in range(F):
        start = hop * fr - W // 2
        seg = np.zeros(dft, np.float32)
        idx = np.arange(start, start + W)
        ok = (idx >= 0) & (idx < x.shape[1])
        seg[:W][ok] = x[0, idx[ok]]
        seg[:W] *= w
        X = np.empty((bins, 2), np.float32)
        O.lib.oracle_r2c(seg, dft, X.reshape(-1))
        m = np.empty(bins, np.float32); f = np.empty(bins, np.float32)
        ref.ref_phase_vocoder_batch(bins, state, np.ascontiguousarray(X[:, 0]), np.ascontiguousarray(X[:, 1]), binf,
                                    np.float32(sr / hop), np.float32(sr), m, f)
        assert np.array_equal(_bits(m), _bits(pv[0, fr, :, 0])), fr
        assert np.array_equal(_bits(f), _bits(pv[0, fr, :, 1])), fr


def test_inverse_phase_vocoder_bit_exact():
    """phase_vocoder.cpp:55-61, including the > pi2 fold and negative-going phases."""
    rng = np.random.default_rng(3)
    n = 400_000
    ar = np.float32(93.75)
    m = (rng.random(n) * 10 ** rng.uniform(-6, 3, n)).astype(np.float32)
    f = rng.uniform(-200, 24100, n).astype(np.float32)
    phase = rng.uniform(-20, 7, n)
    pa, pb = phase.copy(), phase.copy()
    re_r = np.empty(n, np.float32); im_r = np.empty(n, np.float32)
    ref.ref_inverse_phase_vocoder_batch(n, pb, m, f, ar, re_r, im_r)
    rr = C.c_float(); ii = C.c_float(); ph = C.c_double()
    idx = np.arange(0, n, 7)
    re_o = np.empty(n, np.float32); im_o = np.empty(n, np.float32)
    for i in idx:
        ph.value = pa[i]
        O.lib.oracle_inverse_phase_vocoder(C.byref(ph), float(m[i]), float(f[i]), float(ar), C.byref(rr), C.byref(ii))
        re_o[i] = rr.value; im_o[i] = ii.value; pa[i] = ph.value
    assert np.array_equal(_bits(re_o[idx]), _bits(re_r[idx]))
    assert np.array_equal(_bits(im_o[idx]), _bits(im_r[idx]))
    assert np.array_equal(pa[idx], pb[idx])


def test_pvbuffer_unit_conversions_bit_exact():
    """PVBuffer.cpp:356-359,381-384,428-446,526-529"""
    rng = np.random.default_rng(11)
    for sr, hop, dft, W in [(48000.0, 512, 2048, 2048), (48000.0, 128, 4096, 2048), (44100.0, 441, 1024, 1024), (96000.0, 100, 8192, 4096)]:
        ar = np.float32(np.float32(sr) / np.float32(hop))
        fmt = O.RefPVFormat(2, 7, dft // 2 + 1, sr, ar, W)
        assert ref.ref_pv_hop_size(fmt) == O.lib.oracle_hop_size(sr, ar)
        assert ref.ref_pv_dft_size(fmt) == dft
        hop_r = ref.ref_pv_hop_size(fmt)
        for v in rng.uniform(-10, 30000, 500).astype(np.float32):
            v = float(v)
            assert np.float32(ref.ref_pv_bin_to_frequency(fmt, v)).view(np.uint32) == np.float32(O.lib.oracle_bin_to_frequency(v, sr, dft)).view(np.uint32)
            assert np.float32(ref.ref_pv_frequency_to_bin(fmt, v)).view(np.uint32) == np.float32(O.lib.oracle_frequency_to_bin(v, sr, dft)).view(np.uint32)
            assert np.float32(ref.ref_pv_time_to_frame(fmt, v)).view(np.uint32) == np.float32(O.lib.oracle_time_to_frame(v, sr, hop_r)).view(np.uint32)
            assert np.float32(ref.ref_pv_frame_to_time(fmt, v)).view(np.uint32) == np.float32(O.lib.oracle_frame_to_time(v, sr, hop_r)).view(np.uint32)
        # channel -> frame -> bin layout
        assert ref.ref_pv_buffer_pos(fmt, 1, 3, 5) == (1 * 7 + 3) * (dft // 2 + 1) + 5


def test_cubic_spline_bit_exact():
    """the spline restated in oracle/arrange_oracle.cpp against the header the reference vendors (spline/spline.h, compiled unmodified
    into oracle/_ref), the way PV::stretch_spline uses it (PVModify.cpp:427-440): integer knots, float data, evaluated on the
    integers in between -- and on arbitrary points, including outside the knots"""
    rng = np.random.default_rng(77)
    for n in (3, 4, 5, 17, 200, 3000):
        for trial in range(4):
            x = np.concatenate([[0], np.cumsum(rng.integers(1, 9, n - 1))]).astype(np.float64)
            y = rng.normal(0, 100, n).astype(np.float32).astype(np.float64)
            t = np.concatenate([np.arange(0, x[-1] + 1), rng.uniform(-5, x[-1] + 5, 200)])
            ours = O.spline(x, y, t)
            theirs = np.empty(len(t), np.float64)
            ref.ref_spline(x, y, n, np.ascontiguousarray(t), len(t), theirs)
            assert np.array_equal(ours.view(np.uint64), theirs.view(np.uint64)), (n, trial)
    # non-integer knots too
    x = np.sort(rng.uniform(0, 50, 40)); y = rng.normal(0, 1, 40); t = rng.uniform(-1, 51, 500)
    theirs = np.empty(500, np.float64)
    ref.ref_spline(x, y, 40, t, 500, theirs)
    assert np.array_equal(O.spline(x, y, t).view(np.uint64), theirs.view(np.uint64))
