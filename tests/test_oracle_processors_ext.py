"""CPU checks of the oracle's further frame processors (oracle/processors_oracle.cpp).

The reference holds no tests or vectors for these methods and PV/PV.cpp, PV/PVModify.cpp cannot be built here, so what pins
them is: (i) the reference's own Utility/Interpolator.cpp, compiled unmodified into oracle/_ref, for the interpolators;
(ii) known answers that follow from the reference's loops by hand; (iii) independent numpy restatements of the selection /
ranking logic; (iv) the committed golden vectors (regression)."""
import os

import numpy as np
import pytest

import oracle_lib as O

SR = 48000.0
HOP = 256
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "processors", "processors_ext.npz")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def pv():
    return O.analyze(O.noise(2, 12000, seed=5), SR, 512, HOP, 512)       # (2, 47, 257, 2)


def test_interpolators_match_reference_tu():
    ref = O.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    xs = np.concatenate([np.linspace(-2, 3, 20001, dtype=np.float32), np.float32([0, 1, 0.5, 1e-30, -0.0, np.inf])])
    for kind in range(9):
        a = np.array([O.lib.oracle_interpolate(kind, float(x)) for x in xs], np.float32)
        b = np.array([ref.ref_interpolate(kind, float(x)) for x in xs], np.float32)
        assert np.array_equal(_bits(a), _bits(b)), kind


def test_golden_reproduced():
    g = np.load(GOLDEN)
    pv, src, amount = g["pv"], g["src"], g["amount"]
    start, end, Fo = [int(v) for v in g["te_params"]]
    assert np.array_equal(_bits(O.replace_amplitudes(pv, src, amount)), _bits(g["replace"]))
    assert np.array_equal(_bits(O.subtract_amplitudes(pv, src, amount)), _bits(g["subtract"]))
    assert np.array_equal(_bits(O.resonate(pv, SR, 256, 0.05, 0.5)), _bits(g["resonate"]))
    assert np.array_equal(_bits(O.n_loudest_partials(pv, g["n"], False)), _bits(g["retain"]))
    assert np.array_equal(_bits(O.n_loudest_partials(pv, g["n"], True)), _bits(g["remove"]))
    assert np.array_equal(_bits(O.desample(pv, g["ratio"], 0)), _bits(g["desample"]))
    assert np.array_equal(_bits(O.time_extrapolate(pv, SR, start, end, Fo, g["te_samples"])), _bits(g["time_extrapolate"]))


def test_replace_and_subtract_known_answers(pv):
    other = pv[::-1].copy()
    assert np.array_equal(_bits(O.replace_amplitudes(pv, other, 0.0)), _bits(pv))                 # amount 0: unchanged
    full = O.replace_amplitudes(pv, other, 1.0)                                                   # amount 1: source magnitudes, own frequencies
    assert np.array_equal(full[..., 0], other[..., 0] * np.float32(1.0) + pv[..., 0] * np.float32(0.0))
    assert np.array_equal(_bits(full[..., 1]), _bits(pv[..., 1]))
    assert np.array_equal(_bits(O.replace_amplitudes(pv, other, 5.0)), _bits(full))               # clamped (PV.cpp:212)
    part = O.replace_amplitudes(pv, other[:1, :10, :100], 1.0)                                    # outside the overlap: cleared (PV.cpp:215)
    assert not part[1:].any() and not part[:, 10:].any() and not part[:, :, 100:].any()
    assert np.array_equal(_bits(O.subtract_amplitudes(pv, other, 0.0)[..., 0]), _bits(np.abs(pv[..., 0])))
    assert not O.subtract_amplitudes(pv, pv, 1.0)[..., 0].any()                                   # |m - m*1| = 0
    keep = O.subtract_amplitudes(pv, other[:1, :10, :100], 2.5)                                   # outside the overlap: the copy (PV.cpp:246)
    assert np.array_equal(_bits(keep[1:]), _bits(pv[1:]))
    assert np.array_equal(keep[0, :10, :100, 0], np.abs(pv[0, :10, :100, 0] - other[0, :10, :100, 0] * np.float32(2.5)))


def test_resonate_known_answers(pv):
    ch, F, bins, _ = pv.shape
    assert O.lib.oracle_resonate_out_frames(F, 0.0, SR, HOP) == F
    assert O.lib.oracle_resonate_out_frames(F, -3.0, SR, HOP) == F                                # PV.cpp:609-610
    assert O.lib.oracle_resonate_out_frames(F, 0.1, SR, HOP) == F + 19                            # ceil(0.1 * 187.5)
    same = O.resonate(pv, SR, HOP, 0.0, 0.0)                                                      # decay 0: pow(0, dt) = 0 -> every louder-than-0 frame copied
    pos = pv[..., 0] > 0
    assert np.array_equal(_bits(same[pos]), _bits(pv[pos]))
    held = O.resonate(pv, SR, HOP, 0.1, 1.0)                                                      # decay 1: running maximum, then held
    run_max = np.maximum.accumulate(pv[..., 0], axis=1)
    assert np.array_equal(held[:, :F, :, 0], run_max)
    assert np.array_equal(held[:, F:, :, 0], np.repeat(run_max[:, -1:, :], 19, axis=1))
    # frequency of the held tail = frequency of the frame that set the maximum
    arg = np.argmax(pv[..., 0] == run_max[:, -1:, :], axis=1)
    f_of_max = np.take_along_axis(pv[..., 1], arg[:, None, :], axis=1)[:, 0, :]
    assert np.array_equal(held[:, -1, :, 1], f_of_max)
    # geometric tail: one multiplication per frame by the platform powf (PV.cpp:631-632)
    out = O.resonate(pv, SR, HOP, 0.05, 0.5)
    dt = np.float32(1.0) / (np.float32(SR) / np.float32(HOP))
    d = np.float32(np.power(np.float32(0.5), dt, dtype=np.float32))
    assert np.allclose(out[:, F + 3, :, 0], out[:, F + 2, :, 0] * d, rtol=2e-7, atol=0)
    # libm powf vs correctly rounded pow: the two oracle modes stay within the stage tolerance of each other
    rng = np.random.default_rng(0)
    grid = rng.uniform(0, 1, (F + 10, bins)).astype(np.float32)
    a, b = O.resonate(pv, SR, HOP, 0.05, grid, 0), O.resonate(pv, SR, HOP, 0.05, grid, 1)
    assert np.abs(a[..., 0] - b[..., 0]).max() <= 1e-5 * np.abs(a[..., 0]).max()


def test_n_loudest_against_numpy(pv):
    ch, F, bins, _ = pv.shape
    rng = np.random.default_rng(1)
    n = rng.integers(-3, F + 10, F).astype(np.int32)
    retain, remove = O.n_loudest_partials(pv, n, False), O.n_loudest_partials(pv, n, True)
    # the two are complementary and never touch the frequencies (PV.cpp:581-584)
    assert np.array_equal(retain[..., 0] + remove[..., 0], pv[..., 0])
    assert np.array_equal(_bits(retain[..., 1]), _bits(pv[..., 1])) and np.array_equal(_bits(remove[..., 1]), _bits(pv[..., 1]))
    for c in range(ch):
        for f in range(F):
            k = int(np.clip(n[f], 0, F))                                                          # sic: clamped to the number of FRAMES (PV.cpp:556)
            order = np.argsort(-np.abs(pv[c, f, :, 0]), kind="stable")
            expect = np.zeros(bins, np.float32)
            expect[order[:k]] = pv[c, f, order[:k], 0]
            assert np.array_equal(retain[c, f, :, 0], expect), (c, f)


def test_desample_against_numpy(pv):
    ch, F, bins, _ = pv.shape
    ident = O.desample(pv, 1.0, 0)                                                                # ratio 1: every frame selected -> identity but the last frame
    assert np.array_equal(ident[:, :-1, :, 0], (np.float32(1.0) * pv[:, :-1, :, 0]) + np.float32(0.0) * pv[:, 1:, :, 0])
    assert not ident[:, -1].any()                                                                 # PVModify.cpp:453 + :490 (frame < rFrame)
    assert not O.desample(pv, 0.0, 0).any()                                                       # only frame 0 is ever selected (:482)
    # ratio 1/4 (exact in fp32): the accumulator starts at 1 (:461), so frames 0, 3, 7, 11, ... are selected; linear in between
    quarter = O.desample(pv, 0.25, 0)
    sel = np.array([0] + list(range(3, F, 4)))
    last = sel[-1]
    for l, r in zip(sel[:-1], sel[1:]):
        assert np.array_equal(_bits(quarter[:, l, :, 0]), _bits(np.float32(1) * pv[:, l, :, 0] + np.float32(0) * pv[:, r, :, 0]))
    assert not quarter[:, last:].any()
    mid = (np.float32(0.5) * pv[:, 3, :, 0]) + (np.float32(0.5) * pv[:, 7, :, 0])
    assert np.array_equal(quarter[:, 5, :, 0], mid)
    # the frequency is the louder endpoint's (:497)
    w0, w1 = np.float32(0.75) * pv[:, 3, :, 0], np.float32(0.25) * pv[:, 7, :, 0]
    assert np.array_equal(quarter[:, 4, :, 1], np.where(w0 > w1, pv[:, 3, :, 1], pv[:, 7, :, 1]))
    # floor / ceil interpolators hold the left / right endpoint
    assert np.array_equal(O.desample(pv, 0.25, 3)[:, 4, :, 0], np.float32(1) * pv[:, 3, :, 0] + np.float32(0) * pv[:, 7, :, 0])
    assert np.array_equal(O.desample(pv, 0.25, 4)[:, 4, :, 0], np.float32(0) * pv[:, 3, :, 0] + np.float32(1) * pv[:, 7, :, 0])


def test_time_extrapolate_known_answers(pv):
    ch, F, bins, _ = pv.shape
    start, end, Fo = 5, 25, 40
    samples = O.time_extrapolate_interp_samples(start, end, Fo, 0)
    # PVModify.cpp:631-633: the table is indexed from 0 but evaluated at (frame - start_frame)
    assert samples[0] == np.float32(-start) / np.float32(end - start) and samples.shape == (Fo - start,)
    out = O.time_extrapolate(pv, SR, start, end, Fo, samples)
    assert np.array_equal(_bits(out[:, :start]), _bits(pv[:, :start]))                            # :638
    # mix == 1: every candidate is the right frame's MF itself; a bin whose frequency sits at or above its centre lands on its
    # own bin (the float -> Bin truncations of :654 and :658 cancel), so such a bin can only hold itself or a louder neighbour
    k = np.where(samples == 1.0)[0]
    assert k.size == 1
    frame = start + int(k[0])
    right = pv[:, end]
    got = out[:, frame]
    dft = (bins - 1) * 2
    fbin = right[..., 1] / (np.float32(SR) / np.float32(dft))
    own = (fbin >= np.arange(bins, dtype=np.float32)) & (fbin < np.arange(1, bins + 1, dtype=np.float32))
    assert own.mean() > 0.3
    assert (got[..., 0][own] >= np.abs(right[..., 0])[own]).all()
    for c in range(ch):
        placed = got[c][got[c, :, 0] > 0]
        assert np.isin(placed[:, 0], np.abs(right[c, :, 0])).all() and np.isin(placed[:, 1], right[c, :, 1]).all()
    # every output magnitude is one of the candidates' (placement never blends)
    m = samples[10]
    cand = np.abs((np.float32(1) - m) * pv[:, start, :, 0] + m * pv[:, end, :, 0])
    o = out[:, start + 10, :, 0]
    for c in range(ch):
        assert np.isin(o[c][o[c] > 0], cand[c]).all()
