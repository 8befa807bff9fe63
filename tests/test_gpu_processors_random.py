"""Seeded differential sweep of every frame processor against the oracle over random PV shapes (1-3 channels, 1-90 frames, 33-1025
bins, sparse / negative / huge magnitudes) and random user grids.  Bit equality throughout (resonate with a decay grid: the
correctly rounded pow, see oracle_resonate)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SR = 48000.0


@pytest.fixture(scope="module")
def fa():
    import flan_amd
    assert flan_amd.lib.flanhip_device_count() > 0
    return flan_amd


def random_pv(rng):
    ch = int(rng.integers(1, 4))
    F = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65, 90]))
    bins = int(rng.choice([33, 65, 129, 257, 513, 1025]))
    sr_bin = SR / ((bins - 1) * 2)
    pv = np.empty((ch, F, bins, 2), np.float32)
    pv[..., 0] = rng.gamma(0.7, 2.0, (ch, F, bins))
    pv[..., 1] = (np.arange(bins) * sr_bin)[None, None, :] + rng.normal(0, 1.5 * sr_bin, (ch, F, bins))
    kind = rng.integers(0, 4)
    if kind == 0:
        pv[..., 0] *= rng.uniform(0, 1, (ch, F, bins)) < 0.3                    # sparse
    elif kind == 1:
        pv[..., 0] = np.round(pv[..., 0])                                       # quantised: ties, zeros
    elif kind == 2:
        pv[..., 0] *= rng.choice([-1.0, 1.0], (ch, F, bins))                    # negative magnitudes (PV arithmetic can produce them)
    return np.ascontiguousarray(pv, np.float32)


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("seed", range(24))
def test_random_processors(fa, seed):
    rng = np.random.default_rng(1000 + seed)
    pv = random_pv(rng)
    ch, F, bins, _ = pv.shape
    hop = int(rng.choice([64, 128, 256, 512]))
    hop_s = hop / SR
    tag = "seed %d: %s hop %d" % (seed, pv.shape, hop)

    # ---- modify_time: monotone (stretch) and arbitrary maps
    factor = rng.uniform(0.1, 3.0, (F, bins)).astype(np.float32)
    mod = O.stretch_map(factor, SR, hop)
    assert same_bits(fa.modify_time(pv, SR, hop, mod), O.modify_time(pv, SR, hop, mod)), tag + " stretch"
    wild = (rng.uniform(-2, F + 4, (F, bins)) * hop_s).astype(np.float32)
    ref = O.modify_time(pv, SR, hop, wild)
    if ref.size:
        assert same_bits(fa.modify_time(pv, SR, hop, wild), ref), tag + " modify_time"

    # ---- repitch (fused call) and modify_frequency with an arbitrary map
    fgrid = rng.uniform(-0.5, 2.5, (F, bins)).astype(np.float32)
    assert same_bits(fa.repitch(pv, SR, fgrid), O.repitch(pv, SR, fgrid)), tag + " repitch"
    mod_hz = rng.uniform(-500, SR / 2 + 500, (F, bins)).astype(np.float32)
    inmod = rng.uniform(0, SR / 2, (ch, F, bins)).astype(np.float32)
    assert same_bits(fa.modify_frequency(pv, SR, mod_hz, inmod), O.modify_frequency(pv, SR, mod_hz, inmod)), tag + " modify_frequency"

    # ---- shape, with and without shift alignment
    a, b, c, d = [float(v) for v in rng.uniform(-2, 2, 4)]
    for align in (False, True):
        assert same_bits(fa.shape_affine(pv, SR, a, b, c, d * 100, align), O.shape_affine(pv, SR, a, b, c, d * 100, align)), tag + " shape"

    # ---- amplitudes
    other = random_pv(rng)
    amount = rng.uniform(-0.5, 1.5, (F, bins)).astype(np.float32)
    assert same_bits(fa.replace_amplitudes(pv, other, amount), O.replace_amplitudes(pv, other, amount)), tag + " replace"
    assert same_bits(fa.subtract_amplitudes(pv, other, amount), O.subtract_amplitudes(pv, other, amount)), tag + " subtract"

    # ---- resonate
    length = float(rng.uniform(0, 0.05))
    assert same_bits(fa.resonate(pv, SR, hop, length, 0.8), O.resonate(pv, SR, hop, length, 0.8, pow_mode=0)), tag + " resonate const"
    Fo = int(O.lib.oracle_resonate_out_frames(F, length, SR, hop))
    decay = rng.uniform(-0.1, 1.1, (Fo, bins)).astype(np.float32)
    got, ref = fa.resonate(pv, SR, hop, length, decay), O.resonate(pv, SR, hop, length, decay, pow_mode=1)
    assert got.shape == ref.shape and np.mean(got.view(np.uint32) == ref.view(np.uint32)) >= 0.9999, tag + " resonate grid"

    # ---- n loudest
    n = rng.integers(-2, bins + 3, F).astype(np.int32)
    for remove in (False, True):
        assert same_bits(fa.n_loudest_partials(pv, n, remove), O.n_loudest_partials(pv, n, remove)), tag + " n_loudest"

    # ---- desample
    ratio = rng.uniform(-0.1, 1.1, (F, bins)).astype(np.float32)
    interp = int(rng.integers(0, 8))
    assert same_bits(fa.desample(pv, ratio, interp), O.desample(pv, ratio, interp)), tag + " desample"

    # ---- time_extrapolate
    if F >= 3:
        start = int(rng.integers(0, F - 2))
        end = int(rng.integers(start + 1, F))
        out_frames = end + int(rng.integers(1, 40))
        samples = O.time_extrapolate_interp_samples(start, end, out_frames, int(rng.integers(0, 8)))
        assert same_bits(fa.time_extrapolate(pv, SR, start, end, out_frames, samples), O.time_extrapolate(pv, SR, start, end, out_frames, samples)), tag + " time_extrapolate"


@pytest.mark.parametrize("seed", range(16))
def test_random_arranging_processors(fa, seed):
    """the frame-selecting / warping methods (oracle/arrange_oracle.cpp) over the same random shapes: bit equality"""
    rng = np.random.default_rng(5000 + seed)
    pv = random_pv(rng)
    ch, F, bins, _ = pv.shape
    dft = (bins - 1) * 2
    hop = int(rng.choice([64, 128, 256, 512]))
    hop_s = hop / SR
    tag = "seed %d: %s hop %d" % (seed, pv.shape, hop)
    t = (np.arange(F, dtype=np.float32) / np.float32(SR / hop))[:, None] * np.ones((1, bins), np.float32)
    f = (np.arange(bins, dtype=np.float32) * np.float32(SR) / np.float32(dft))[None, :] * np.ones((F, 1), np.float32)

    # ---- get_frame, freeze, cut_frames, join
    pos = float(np.float32(rng.uniform(0, F - 1)))
    interp = int(rng.integers(0, 7))
    assert same_bits(fa.get_frame(pv, pos, interp), O.get_frame(pv, pos, interp)), tag + " get_frame"
    n_ev = int(rng.integers(0, 6))
    times = rng.uniform(-hop_s, (F + 1) * hop_s, n_ev).astype(np.float32)
    lengths = rng.uniform(-hop_s, 6 * hop_s, n_ev).astype(np.float32)
    assert same_bits(fa.freeze(pv, SR, hop, times, lengths), O.freeze(pv, SR, hop, times, lengths)), tag + " freeze"
    a, b = sorted(int(v) for v in rng.integers(-3, F + 3, 2))
    got, ref = fa.cut_frames(pv, a, b), O.cut_frames(pv, a, b)
    assert (got is None and ref is None) or same_bits(got, ref), tag + " cut_frames"
    other = random_pv(rng)
    assert same_bits(fa.join([pv, other, pv[:1]]), O.join([pv, other, pv[:1]])), tag + " join"

    # ---- select with a random selector grid
    Fo = int(rng.integers(1, 2 * F + 3))
    sel = np.empty((Fo, bins, 2), np.float32)
    sel[..., 0] = rng.uniform(-2 * hop_s, (F + 2) * hop_s, (Fo, bins))
    sel[..., 1] = rng.uniform(-500.0, SR / 2 + 500.0, (Fo, bins))
    assert same_bits(fa.select(pv, SR, hop, sel), O.select(pv, SR, hop, sel)), tag + " select"

    # ---- add_octaves / add_harmonics with random series
    for mode, H in ((0, int(rng.integers(1, 20))), (1, int(rng.integers(1, bins + 1)))):
        series = rng.uniform(-0.5, 1.5, (F, H)).astype(np.float32)
        assert same_bits(fa.harmonic_scale(pv, SR, series, mode), O.harmonic_scale(pv, SR, series, mode)), tag + " harmonic_scale %d" % mode

    # ---- smear_time with random grids
    smear = (rng.uniform(-1, 5, (F, bins)) * hop_s).astype(np.float32) if rng.integers(0, 2) else float(rng.uniform(0, 4) * hop_s)
    gran = rng.integers(-1, 4, (F, bins)).astype(np.int32) if rng.integers(0, 2) else int(rng.integers(1, 4))
    left, Fs, half = O.smear_time_plan(F, bins, SR, hop, smear)
    assert (left, Fs, half) == fa.smear_time_plan(F, bins, SR, hop, smear), tag + " smear plan"
    if Fs > 0:
        dist = rng.uniform(0, 1, 2 * half).astype(np.float32)
        assert same_bits(fa.smear_time(pv, SR, hop, smear, gran, dist, left, Fs), O.smear_time(pv, SR, hop, smear, gran, dist, left, Fs)), tag + " smear_time"

    # ---- modify with a smooth random warp (jittered affine map: quads stay small)
    if F >= 2:
        a11, a22 = rng.uniform(0.5, 1.8), rng.uniform(0.6, 1.4)
        grid = np.stack([t * np.float32(a11) + np.float32(rng.uniform(0, 3) * hop_s) + rng.uniform(-0.3, 0.3, t.shape).astype(np.float32) * np.float32(hop_s),
                         f * np.float32(a22) + np.float32(rng.uniform(-200, 200)) + rng.uniform(-0.3, 0.3, f.shape).astype(np.float32) * np.float32(SR / dft)], -1).astype(np.float32)
        in_f = rng.uniform(0, SR / 2, (ch, F, bins)).astype(np.float32)
        Fm = O.modify_out_frames(grid, SR, hop)
        assert Fm == fa.modify_out_frames(grid, SR, hop), tag + " modify frames"
        if Fm > 0:
            k = int(rng.choice([0, 1, 3, 4, 5, 6]))
            assert same_bits(fa.modify(pv, SR, hop, grid, in_f, k, Fm), O.modify(pv, SR, hop, grid, in_f, k, Fm)), tag + " modify interp %d" % k

    # ---- stretch_spline with random steps
    if F >= 3:
        steps = rng.integers(1, 6, F - 1).astype(np.uint32)
        assert same_bits(fa.stretch_spline(pv, steps), O.stretch_spline(pv, steps)), tag + " stretch_spline"
