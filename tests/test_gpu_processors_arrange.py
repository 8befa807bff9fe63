"""GPU parity of the frame-selecting / re-placing PV methods through the C ABI: get_frame, select, freeze, cut_frames, join,
add_octaves / add_harmonics.  Copies, gathers and plain fp32 with the reference's operation order kept: the bar is BIT EQUALITY
with the oracle (oracle/arrange_oracle.cpp)."""
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
    return O.analyze(O.noise(2, 30000, seed=77), SR, 1024, HOP, 1024)          # (2, 118, 513, 2)


@pytest.fixture(scope="module")
def pv_wide():
    return O.analyze(O.noise(3, 24000, seed=78), SR, 1024, HOP, 2048)          # (3, 94, 1025, 2)


def assert_identical(name, got, ref):
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    same = float(np.mean(got.view(np.uint32) == ref.view(np.uint32)))
    print("\n[P4 %s] bit-identical=%.6f" % (name, same))
    assert same == 1.0, name


def test_get_frame(fa, pv_small, pv_wide):
    for pv in (pv_small, pv_wide):
        F = pv.shape[1]
        for pos in (0.0, 1.0, 17.25, 60.5, F - 1.0, F - 1.75):
            for interp in range(9):
                assert_identical("get_frame pos=%g interp=%d" % (pos, interp), fa.get_frame(pv, pos, interp), O.get_frame(pv, pos, interp))
    with pytest.raises(fa.FlanHipError):
        fa.get_frame(pv_small, float(pv_small.shape[1]))                            # the caller clamps (PV.cpp:28); beyond the PV is an error
    with pytest.raises(fa.FlanHipError):
        fa.get_frame(pv_small, -0.5)


def test_freeze(fa, pv_small, pv_wide):
    rng = np.random.default_rng(4)
    for pv in (pv_small, pv_wide):
        F = pv.shape[1]
        T = F * HOP / SR
        cases = [([], []), ([0.1], [0.05]), ([0.1], [0.0]), ([0.0, T, 0.1, 0.1], [0.01, 0.02, 0.03, 0.3]), ([-1.0, 1e9], [0.02, -1.0])]
        cases.append((rng.uniform(0, T, 20), rng.uniform(0, 0.05, 20)))
        for times, lengths in cases:
            assert_identical("freeze %d events" % len(times), fa.freeze(pv, SR, HOP, times, lengths), O.freeze(pv, SR, HOP, times, lengths))


def test_cut_frames_and_join(fa, pv_small, pv_wide):
    F = pv_small.shape[1]
    for start, end in ((0, F), (3, 11), (-5, 40), (F - 2, F + 7), (50, 51)):
        assert_identical("cut_frames %d:%d" % (start, end), fa.cut_frames(pv_small, start, end), O.cut_frames(pv_small, start, end))
    for start, end in ((5, 5), (9, 2), (F - 1, F + 7), (F + 3, F + 9)):
        assert fa.cut_frames(pv_small, start, end) is None and O.cut_frames(pv_small, start, end) is None
    parts = [pv_small[:, :7], pv_small[:, 7:20], pv_small[:, 20:]]
    assert_identical("join of cuts", fa.join(parts), pv_small)
    assert_identical("join, other shapes", fa.join([pv_small, pv_wide, pv_small[:1, :9, :100]]), O.join([pv_small, pv_wide, pv_small[:1, :9, :100]]))
    assert_identical("join, wide first", fa.join([pv_wide, pv_small]), O.join([pv_wide, pv_small]))


def selector_grids(pv, Fo, seed):
    ch, F, bins, _ = pv.shape
    rng = np.random.default_rng(seed)
    T = F * HOP / SR
    sel = np.empty((Fo, bins, 2), np.float32)
    sel[..., 0] = rng.uniform(-0.02, T * 1.1, (Fo, bins))
    sel[..., 1] = rng.uniform(-200.0, SR / 2 * 1.1, (Fo, bins))
    yield "random", sel
    # the identity selector: every output point reads its own time / frequency
    t = (np.arange(Fo, dtype=np.float32) / np.float32(SR / HOP))[:, None]
    f = (np.arange(bins, dtype=np.float32) * np.float32(SR) / np.float32((bins - 1) * 2))[None, :]
    ident = np.stack(np.broadcast_arrays(t, f), axis=-1).astype(np.float32)
    yield "identity", ident
    weird = sel.copy()
    weird[0, :5] = [(np.nan, 100.0), (0.01, np.nan), (np.inf, 100.0), (0.01, -np.inf), (0.01, 0.5)]
    weird[-1, 5:8] = [(1e30, 1e30), (-1e30, 5.0), (0.0, 0.0)]
    yield "non-finite", weird


def test_select(fa, pv_small, pv_wide):
    for pv in (pv_small, pv_wide):
        for Fo in (1, 37, pv.shape[1] + 20):
            for name, sel in selector_grids(pv, Fo, seed=Fo):
                assert_identical("select/%s Fo=%d" % (name, Fo), fa.select(pv, SR, HOP, sel), O.select(pv, SR, HOP, sel))


def test_add_octaves_and_harmonics(fa, pv_small, pv_wide):
    rng = np.random.default_rng(6)
    for pv in (pv_small, pv_wide):
        ch, F, bins, _ = pv.shape
        H_oct = int(np.ceil(np.log2(np.float32(bins * np.float32(SR) / np.float32((bins - 1) * 2)))))
        for mode, H in ((0, H_oct), (1, bins), (1, 7), (0, 40), (1, 0)):
            for sname, series in (("ones", np.ones((F, H), np.float32)), ("random", rng.uniform(-0.3, 1.0, (F, H)).astype(np.float32)),
                                  ("decay", np.tile((0.7 ** np.arange(H, dtype=np.float32))[None, :], (F, 1)).astype(np.float32))):
                assert_identical("harmonic_scale mode=%d H=%d %s" % (mode, H, sname), fa.harmonic_scale(pv, SR, series, mode), O.harmonic_scale(pv, SR, series, mode))
    # dft 8192: the input row and the series no longer fit next to the placement keys (the kernel's other path); dft 16384: refused
    big = O.analyze(O.noise(1, 4000, seed=3), SR, 1024, HOP, 8192)
    for mode, H in ((0, 15), (1, big.shape[2])):
        series = rng.uniform(0.0, 1.0, (big.shape[1], H)).astype(np.float32)
        assert_identical("harmonic_scale dft 8192 mode=%d" % mode, fa.harmonic_scale(big, SR, series, mode), O.harmonic_scale(big, SR, series, mode))
    huge = np.zeros((1, 2, 8193, 2), np.float32)
    with pytest.raises(fa.FlanHipError):
        fa.harmonic_scale(huge, SR, np.ones((2, 4), np.float32), 0)
    # ties, zeros, negative and non-finite magnitudes / frequencies in the source
    pv = pv_small.copy()
    pv[0, 3, :, 0] = 1.0                                                             # every candidate equally loud: the first visited wins
    pv[0, 4, 10:20, 0] = 0.0
    pv[0, 5, 10:20, 0] = -2.0
    pv[0, 6, 10, 0] = np.inf
    pv[0, 7, 10:14, 1] = [np.nan, np.inf, -5.0, 1.0]
    pv[1, 8, :, 1] = 30.0                                                            # every bin claims the same overtone bins
    for mode, H in ((0, 15), (1, pv.shape[2])):
        series = np.ones((pv.shape[1], H), np.float32)
        assert_identical("harmonic_scale special mode=%d" % mode, fa.harmonic_scale(pv, SR, series, mode), O.harmonic_scale(pv, SR, series, mode))


def test_smear_time(fa, pv_small, pv_wide):
    rng = np.random.default_rng(21)
    for pv in (pv_small, pv_wide):
        ch, F, bins, _ = pv.shape
        grids = [(0.02, 1), (0.1, 5), (0.0, 5), (0.003, 1),
                 (rng.uniform(-0.02, 0.12, (F, bins)).astype(np.float32), rng.integers(-2, 7, (F, bins)).astype(np.int32)),
                 (rng.uniform(0.0, 0.05, (F, bins)).astype(np.float32), 2)]
        weird = rng.uniform(0.0, 0.05, (F, bins)).astype(np.float32)
        weird[3, :4] = [np.nan, np.inf, -np.inf, 1e-30]
        weird[3, 1] = 0.2                                                            # (an infinite size is refused by the plan: not a frame count)
        grids.append((weird, 1))
        for smear, gran in grids:
            left, Fo, half = O.smear_time_plan(F, bins, SR, HOP, smear)
            assert (left, Fo, half) == fa.smear_time_plan(F, bins, SR, HOP, smear)
            for dname, dist in (("hann", O.smear_distribution(half)), ("box", np.ones(2 * half, np.float32)), ("signed", rng.uniform(-1, 1, 2 * half).astype(np.float32))):
                got = fa.smear_time(pv, SR, HOP, smear, gran, dist, left, Fo)
                ref = O.smear_time(pv, SR, HOP, smear, gran, dist, left, Fo)
                assert_identical("smear_time %s / %s" % ("grid" if not np.isscalar(smear) else smear, dname), got, ref)


def warp_grids(pv, hop, seed):
    ch, F, bins, _ = pv.shape
    dft = (bins - 1) * 2
    rng = np.random.default_rng(seed)
    t = (np.arange(F, dtype=np.float32) / np.float32(SR / hop))[:, None] * np.ones((1, bins), np.float32)
    f = (np.arange(bins, dtype=np.float32) * np.float32(SR) / np.float32(dft))[None, :] * np.ones((F, 1), np.float32)
    T = np.float32(F * hop / SR)
    yield "identity", np.stack([t, f], -1)
    yield "stretch 1.7, transpose 1.3", np.stack([t * np.float32(1.7), f * np.float32(1.3)], -1)
    yield "shrink", np.stack([t * np.float32(0.4) + np.float32(0.01), f * np.float32(0.6) + np.float32(50)], -1)
    yield "bend", np.stack([t + np.float32(0.02) * np.sin(f / np.float32(3000)).astype(np.float32), f * (np.float32(1) + np.float32(0.3) * t / T) + np.float32(40)], -1)
    yield "fold", np.stack([T * np.abs(np.sin(t / T * np.float32(5))).astype(np.float32), np.float32(SR / 2) * np.abs(np.cos(f / np.float32(7000))).astype(np.float32)], -1)
    yield "shear", np.stack([t + f / np.float32(SR) * T * np.float32(0.2), f + t / T * np.float32(2000)], -1)
    jitter = np.stack([t + rng.uniform(-0.4, 0.4, t.shape).astype(np.float32) * np.float32(hop / SR),
                       f + rng.uniform(-0.4, 0.4, f.shape).astype(np.float32) * np.float32(SR / dft)], -1)
    yield "jitter", jitter
    holes = jitter.copy()
    holes[5, 7] = (np.nan, 100.0); holes[6, 9] = (0.01, np.inf); holes[8, 3] = (-np.inf, np.nan); holes[F - 1, bins - 1] = (np.nan, np.nan)
    yield "non-finite corners", holes
    yield "partly before time zero", np.stack([t - np.float32(0.05), f - np.float32(900)], -1)


def test_modify(fa, pv_small, pv_wide):
    for pv, hop in ((pv_small, HOP), (pv_wide, HOP)):
        ch, F, bins, _ = pv.shape
        in_f = np.random.default_rng(9).uniform(0, 24000, (ch, F, bins)).astype(np.float32)
        for name, grid in warp_grids(pv, hop, seed=F):
            grid = np.ascontiguousarray(grid, np.float32)
            Fo = O.modify_out_frames(grid, SR, hop)
            assert Fo == fa.modify_out_frames(grid, SR, hop), name
            if Fo <= 0:
                continue
            for interp in ((0, 5, 6, 1, 3, 4, 2) if name in ("bend", "jitter") else (0,)):
                got = fa.modify(pv, SR, hop, grid, in_f, interp, Fo)
                ref = O.modify(pv, SR, hop, grid, in_f, interp, Fo)
                assert_identical("modify/%s interp=%d Fo=%d" % (name, interp, Fo), got, ref)
                assert name == "partly before time zero" or got.any()
    # equal magnitudes everywhere: every candidate ties, the first quad in ( frame, bin ) order gives the frequency
    flat = pv_small.copy(); flat[..., 0] = 1.0
    in_f = np.random.default_rng(10).uniform(0, 24000, flat.shape[:3]).astype(np.float32)
    for name, grid in warp_grids(flat, HOP, seed=1):
        if name in ("shrink", "fold", "jitter"):
            grid = np.ascontiguousarray(grid, np.float32)
            Fo = O.modify_out_frames(grid, SR, HOP)
            assert_identical("modify ties/%s" % name, fa.modify(flat, SR, HOP, grid, in_f, 0, Fo), O.modify(flat, SR, HOP, grid, in_f, 0, Fo))


def test_stretch_spline(fa, pv_small, pv_wide):
    """fp64 recurrences in the vendored spline's operation order: bit-identical to the checker, which is itself pinned bit for bit
    against the real spline.h (tests/test_oracle_vs_ref.py)"""
    rng = np.random.default_rng(31)
    for pv in (pv_small, pv_wide, pv_small[:, :3].copy(), pv_wide[:1, :4].copy()):
        F = pv.shape[1]
        for name, steps in (("ones", np.ones(F - 1, np.uint32)), ("twos", np.full(F - 1, 2, np.uint32)), ("random 1..9", rng.integers(1, 10, F - 1).astype(np.uint32)),
                            ("one long", np.concatenate([[37], np.ones(F - 2)]).astype(np.uint32))):
            got, ref = fa.stretch_spline(pv, steps), O.stretch_spline(pv, steps)
            assert_identical("stretch_spline/%s F=%d" % (name, F), got, ref)
    # steps of one: every output frame but the first sits ON a knot and is evaluated from the segment before it (lower_bound,
    # spline.h:380-381): equal to the input up to the rounding of that evaluation, not bit for bit
    ones = fa.stretch_spline(pv_small, np.ones(pv_small.shape[1] - 1, np.uint32))
    assert ones.shape[1] == pv_small.shape[1] - 1
    assert np.array_equal(ones[:, 0], pv_small[:, 0])
    assert np.allclose(ones, pv_small[:, :-1], rtol=1e-5, atol=1e-3)
    with pytest.raises(fa.FlanHipError):
        fa.stretch_spline(pv_small[:, :2].copy(), np.ones(1, np.uint32))           # fewer than three knots
    with pytest.raises(fa.FlanHipError):
        fa.stretch_spline(pv_small, np.zeros(pv_small.shape[1] - 1, np.uint32))    # a step of 0: the knots would not increase


def test_golden_fixture(fa):
    """the committed vectors of tests/golden/processors/processors_arrange.npz through the GPU library: bit for bit"""
    from test_oracle_processors_arrange import _golden, golden_outputs
    g = _golden()
    for name, got in golden_outputs(fa, g).items():
        assert_identical("golden/" + name, got, g[name])
