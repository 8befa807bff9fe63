"""GPU parity of the further PV frame processors through the C ABI: replace_amplitudes, subtract_amplitudes,
resonate, retain/remove_n_loudest_partials, desample, time_extrapolate.  These are integer/selection and plain fp32 work with
the reference's operation order kept, so the bar is BIT EQUALITY with the oracle -- except resonate with a sampled decay
grid, where pow() is libm specific (see oracle/processors_oracle.cpp oracle_resonate)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

SR = 48000.0
HOP = 256


@pytest.fixture(scope="module")
def fa():
    import flan_amd
    assert flan_amd.lib.flanhip_device_count() > 0
    return flan_amd


@pytest.fixture(scope="module")
def pv_small():
    x = O.noise(2, 30000, seed=77)
    return O.analyze(x, SR, 1024, HOP, 1024)          # (2, 118, 513, 2)


@pytest.fixture(scope="module")
def pv_other():
    x = O.noise(3, 24000, seed=78)
    return O.analyze(x, SR, 1024, HOP, 2048)          # (3, 94, 1025, 2): other channel / frame / bin counts


def bits_equal(got, ref):
    return float(np.mean(got.view(np.uint32) == ref.view(np.uint32)))


def assert_identical(name, got, ref):
    assert got.shape == ref.shape, name
    same = bits_equal(got, ref)
    print("\n[P4 %s] bit-identical=%.6f" % (name, same))
    assert same == 1.0, name


def amount_grids(F, bins):
    rng = np.random.default_rng(11)
    yield "const0.3", 0.3
    yield "const1.7", 1.7                               # beyond the clamp of replace_amplitudes
    yield "random", rng.uniform(-0.5, 1.5, (F, bins)).astype(np.float32)


def test_replace_and_subtract_amplitudes(fa, pv_small, pv_other):
    ch, F, bins, _ = pv_small.shape
    for src_name, src in (("same-shape", pv_small[::-1].copy()), ("other-shape", pv_other), ("fewer", pv_small[:1, :50, :100].copy())):
        for name, g in amount_grids(F, bins):
            assert_identical("replace/%s/%s" % (src_name, name), fa.replace_amplitudes(pv_small, src, g), O.replace_amplitudes(pv_small, src, g))
            assert_identical("subtract/%s/%s" % (src_name, name), fa.subtract_amplitudes(pv_small, src, g), O.subtract_amplitudes(pv_small, src, g))


def test_resonate_constant_decay(fa, pv_small):
    """constant decay: the library raises it with the host's powf, exactly the reference's call on this platform"""
    ch, F, bins, _ = pv_small.shape
    for length, decay in ((0.0, 0.5), (0.25, 0.5), (0.1, 0.999), (0.3, 0.0), (0.2, 1.0), (0.2, 7.0), (-1.0, 0.2)):
        ref = O.resonate(pv_small, SR, HOP, length, decay, pow_mode=0)
        got = fa.resonate(pv_small, SR, HOP, length, decay)
        assert_identical("resonate/const len=%g decay=%g" % (length, decay), got, ref)


def test_resonate_decay_grid(fa, pv_small):
    ch, F, bins, _ = pv_small.shape
    Fo = int(O.lib.oracle_resonate_out_frames(F, 0.2, SR, HOP))
    rng = np.random.default_rng(3)
    grid = rng.uniform(-0.1, 1.1, (Fo, bins)).astype(np.float32)
    got = fa.resonate(pv_small, SR, HOP, 0.2, grid)
    exact = O.resonate(pv_small, SR, HOP, 0.2, grid, pow_mode=1)   # correctly rounded pow, what the device evaluates
    libm = O.resonate(pv_small, SR, HOP, 0.2, grid, pow_mode=0)    # the platform powf the reference would call
    same = bits_equal(got, exact)
    d = np.abs(got[..., 0].astype(np.float64) - libm[..., 0]).max() / np.abs(libm[..., 0]).max()
    print("\n[P4 resonate/grid] bit-identical to correctly rounded pow: %.6f; max |dm| vs libm powf, relative to max m: %.2e" % (same, d))
    assert same >= 0.9999
    assert d <= 1e-5


def test_n_loudest_partials(fa, pv_small, pv_other):
    for pv in (pv_small, pv_other):
        ch, F, bins, _ = pv.shape
        rng = np.random.default_rng(21)
        cases = [("n=0", 0), ("n=1", 1), ("n=40", 40), ("n=bins-1", bins - 1), ("n>=bins", bins + 5), ("n<0", -3),
                 ("per-frame", rng.integers(-2, min(bins, F) + 4, F).astype(np.int32))]
        for name, n in cases:
            for remove in (False, True):
                ref = O.n_loudest_partials(pv, n, remove)
                got = fa.n_loudest_partials(pv, n, remove)
                assert_identical("n_loudest/%s/%s bins=%d" % (name, "remove" if remove else "retain", bins), got, ref)


def test_n_loudest_partials_ties(fa):
    """equal magnitudes (quantised data, silent frames, negative and zero magnitudes): ranked by ascending bin"""
    rng = np.random.default_rng(4)
    pv = np.zeros((2, 30, 200, 2), np.float32)
    pv[..., 0] = rng.integers(-3, 4, pv.shape[:3]).astype(np.float32)        # many exact ties, signs mixed
    pv[..., 1] = rng.uniform(0, 24000, pv.shape[:3]).astype(np.float32)
    pv[0, 5, :, 0] = 0.0                                                     # a silent frame
    pv[1, 6, :, 0] = 2.0                                                     # a frame of one value
    for n in (1, 17, 64, 65, 130, 199):
        for remove in (False, True):
            assert_identical("n_loudest/ties n=%d" % n, fa.n_loudest_partials(pv, n, remove), O.n_loudest_partials(pv, n, remove))


def test_desample(fa, pv_small):
    ch, F, bins, _ = pv_small.shape
    rng = np.random.default_rng(8)
    ramp = np.linspace(0.02, 1.0, F, dtype=np.float32)[:, None] * np.ones((1, bins), np.float32)
    cases = [("const0.25", 0.25), ("const1", 1.0), ("const0", 0.0), ("const0.013", 0.013), ("random", rng.uniform(-0.2, 1.2, (F, bins)).astype(np.float32)),
             ("ramp", ramp)]
    for name, ratio in cases:
        for interp in (0, 1, 2, 3, 4, 5, 6, 7):                                # every interpolator evaluated exactly on the device
            ref = O.desample(pv_small, ratio, interp)
            got = fa.desample(pv_small, ratio, interp)
            assert_identical("desample/%s/interp%d" % (name, interp), got, ref)
    ref = O.desample(pv_small, 0.1, 8)                                          # sine: cosf is libm specific
    got = fa.desample(pv_small, 0.1, 8)
    assert bits_equal(got[..., 1], ref[..., 1]) >= 0.999
    np.testing.assert_allclose(got[..., 0], ref[..., 0], rtol=1e-5, atol=1e-6)


def test_desample_long(fa):
    """more than one 64-frame tile with a ragged tail, bins not a multiple of 64"""
    rng = np.random.default_rng(2)
    pv = rng.uniform(0, 1, (1, 333, 70, 2)).astype(np.float32)
    ratio = rng.uniform(0, 0.2, (333, 70)).astype(np.float32)
    assert_identical("desample/long", fa.desample(pv, ratio, 0), O.desample(pv, ratio, 0))


def test_resonate_long(fa):
    rng = np.random.default_rng(6)
    pv = rng.uniform(0, 1, (2, 333, 70, 2)).astype(np.float32)
    pv[..., 0] *= rng.uniform(0, 1, (2, 333, 70)) < 0.05                       # sparse onsets, long decays between them
    assert_identical("resonate/long", fa.resonate(pv, SR, HOP, 0.7, 0.9), O.resonate(pv, SR, HOP, 0.7, 0.9, pow_mode=0))


def test_time_extrapolate(fa, pv_small, pv_other):
    for pv in (pv_small, pv_other):
        ch, F, bins, _ = pv.shape
        for (start, end, ext, interp) in ((10, 60, 80, 0), (0, F - 1, 40, 0), (30, 31, 100, 5), (5, 90, 10, 7), (20, 70, 64, 2)):
            Fo = end + ext
            samples = O.time_extrapolate_interp_samples(start, end, Fo, interp)
            ref = O.time_extrapolate(pv, SR, start, end, Fo, samples)
            got = fa.time_extrapolate(pv, SR, start, end, Fo, samples)
            assert_identical("time_extrapolate/%d-%d+%d interp%d bins=%d" % (start, end, ext, interp, bins), got, ref)


def test_time_extrapolate_collisions(fa):
    """quantised frequencies and magnitudes: many candidates per target bin with equal magnitudes (first source bin wins)"""
    rng = np.random.default_rng(12)
    bins, F = 129, 40
    pv = np.zeros((2, F, bins, 2), np.float32)
    pv[..., 0] = rng.integers(0, 4, pv.shape[:3]).astype(np.float32)
    pv[..., 1] = (rng.integers(0, 40, pv.shape[:3]) * 187.5 * 3).astype(np.float32)
    start, end, Fo = 3, 25, 70
    samples = O.time_extrapolate_interp_samples(start, end, Fo, 0)
    assert_identical("time_extrapolate/collisions", fa.time_extrapolate(pv, SR, start, end, Fo, samples), O.time_extrapolate(pv, SR, start, end, Fo, samples))


def test_shape_alignment_collisions(fa):
    """the wavefront-per-row placement of PV::shape with shift alignment on data built to collide"""
    rng = np.random.default_rng(13)
    pv = np.zeros((2, 50, 257, 2), np.float32)
    pv[..., 0] = rng.integers(0, 3, pv.shape[:3]).astype(np.float32)
    pv[..., 1] = (rng.integers(0, 30, pv.shape[:3]) * 93.75 * 8).astype(np.float32)
    for (a, b, c, d) in ((1.0, 0.0, 0.5, 0.0), (1.0, 0.0, 1.0, 300.0), (-1.0, 2.0, 2.0, -100.0)):
        assert_identical("shape/aligned-collisions", fa.shape_affine(pv, SR, a, b, c, d, True), O.shape_affine(pv, SR, a, b, c, d, True))


def test_golden_vectors(fa):
    """tests/golden/processors/processors_ext.npz (written by tests/golden/make_golden.py from the oracle)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "processors", "processors_ext.npz"))
    pv, src, amount = g["pv"], g["src"], g["amount"]
    start, end, Fo = [int(v) for v in g["te_params"]]
    assert_identical("golden/replace", fa.replace_amplitudes(pv, src, amount), g["replace"])
    assert_identical("golden/subtract", fa.subtract_amplitudes(pv, src, amount), g["subtract"])
    assert_identical("golden/resonate", fa.resonate(pv, SR, 256, 0.05, 0.5), g["resonate"])
    assert_identical("golden/retain", fa.n_loudest_partials(pv, g["n"], False), g["retain"])
    assert_identical("golden/remove", fa.n_loudest_partials(pv, g["n"], True), g["remove"])
    assert_identical("golden/desample", fa.desample(pv, g["ratio"], 0), g["desample"])
    assert_identical("golden/time_extrapolate", fa.time_extrapolate(pv, SR, start, end, Fo, g["te_samples"]), g["time_extrapolate"])


def test_n_loudest_partials_wide_rows(fa):
    """rows too wide for the register-resident keys (dft 8192: 4097 bins) take the LDS variant; dft 4096 the widest register one"""
    rng = np.random.default_rng(17)
    for bins in (2049, 4097):
        pv = rng.uniform(0, 1, (1, 7, bins, 2)).astype(np.float32)
        n = np.array([0, 1, 5, 6, 7, 3, 2], np.int32)
        for remove in (False, True):
            assert_identical("n_loudest/wide bins=%d" % bins, fa.n_loudest_partials(pv, n, remove), O.n_loudest_partials(pv, n, remove))
