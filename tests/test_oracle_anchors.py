"""Known-answer anchors for the frame loops the reference cannot be built for here (Conversions/AudioPV.cpp needs
FFTW3f + libsndfile; PV/PVModify.cpp needs MSVC's std::_Pi).  The numbers are the ones SURVEY.md section 8c records from
the reference's own translation units run during the survey (mono 5 s, x[n] = 0.5 sin(2 pi 440 n / 48000) computed
in double then rounded, window 2048, hop 512)."""
import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def sine_pv():
    x = O.sine(240000)
    return x, O.analyze(x, 48000.0, 2048, 512, 2048)


def test_shapes(sine_pv):
    x, pv = sine_pv
    assert pv.shape == (1, 469, 1025, 2)          # F = 240000/512 + 1 (integer division, AudioPV.cpp:17)


def test_hann_anchors():
    w = O.hann_window(2048)
    assert w[1] == np.float32(2.3553948e-06)
    assert w[1023] == np.float32(0.9999994)
    assert w[0] == 0.0 and 0.0 < w[2047] < 1e-13   # symmetric (not periodic) Hann; float 2*pi leaves 7.7e-15 at the far end


@pytest.mark.parametrize("frame,bin,m,f", [
    (100, 18, 171.278, 440.000), (100, 19, 247.517, 440.000), (100, 20, 86.0653, 440.000),
    (0, 19, 126.01691, 489.28607), (1, 19, 226.42621, 442.31033), (468, 19, 208.68384, 441.01926),
    (100, 0, 0.014684752, -46.875), (100, 1024, 3.3868e-07, 24046.875)])
def test_analysis_anchors(sine_pv, frame, bin, m, f):
    _, pv = sine_pv
    assert pv[0, frame, bin, 0] == pytest.approx(m, rel=2e-5)
    assert pv[0, frame, bin, 1] == pytest.approx(f, rel=2e-6)


def test_analysis_energy(sine_pv):
    _, pv = sine_pv
    assert np.sum(pv[..., 0].astype(np.float64) ** 2) == pytest.approx(4.6019687e7, rel=1e-7)


def test_synthesis_anchors(sine_pv):
    x, pv = sine_pv
    out, flag = O.synthesize(pv, 48000.0, 48000.0 / 512, 2048)
    assert flag == 0
    assert out.shape == (1, 240128)                # F * hop >= input length (AudioPV.cpp:93)
    assert out[0, 1000] == pytest.approx(0.43333316, rel=2e-7)
    assert x[0, 1000] == pytest.approx(0.43301269, rel=2e-7)   # round-trip gain 1.00074 (the "2.67" constant, :99)
    assert out[0, 120000] == pytest.approx(-3.957e-06, rel=2e-3)
    assert out[0, 239999] == pytest.approx(-0.0092202881, rel=2e-6)
    assert np.sum(out.astype(np.float64) ** 2) == pytest.approx(30006.053, rel=2e-8)


def test_dft4096_anchors():
    x = O.sine(240000)
    pv = O.analyze(x, 48000.0, 2048, 512, 4096)
    assert pv.shape == (1, 469, 2049, 2)
    assert np.sum(pv[..., 0].astype(np.float64) ** 2) == pytest.approx(9.20393091e7, rel=1e-8)
    out, _ = O.synthesize(pv, 48000.0, 48000.0 / 512, 2048)
    assert out[0, 1000] == pytest.approx(0.43333319, rel=2e-7)
    assert np.sum(out.astype(np.float64) ** 2) == pytest.approx(30006.0548, rel=2e-9)


def test_stretch_repitch_anchors():
    """stretch(lambda->2) maps 469 -> 938 frames; repitch(lambda->2) maps 1187.82 Hz -> 2422.51 Hz
    (f -> 2f + 2 bin widths: the inclusive prefix sum is part of the semantics)."""
    x = O.noise(1, 240000, seed=1234)
    pv = O.analyze(x, 48000.0, 2048, 512, 2048)
    F, bins = pv.shape[1], pv.shape[2]
    two = np.full((F, bins), 2.0, np.float32)
    st = O.stretch(pv, 48000.0, 512, two)
    assert st.shape[1] == 938
    g, inmod = O.repitch_map(pv, 48000.0, two)
    # a partial at 1187.82 Hz lands on 2*1187.82 + 2*23.4375 = 2422.51
    probe = pv.copy()
    probe[0, 0, 50, 1] = 1187.82
    _, inmod = O.repitch_map(probe, 48000.0, two)
    assert inmod[0, 0, 50] == pytest.approx(2422.51, abs=0.01)
