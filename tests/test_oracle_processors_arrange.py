"""The CPU checker for the frame-selecting / re-placing PV methods (oracle/arrange_oracle.cpp) against independent numpy
statements of the same reference loops (PV/PV.cpp:24-39, :92-198, :362-419, :643-720), and the library's host-side planning
helpers (pure arithmetic, no device) against the checker."""
import numpy as np

import oracle_lib as O

SR = 48000.0
HOP = 256


def small_pv(ch=2, n=9000, dft=512, seed=5):
    return O.analyze(O.noise(ch, n, seed=seed), SR, dft, HOP, dft)


def test_get_frame_is_the_blend_of_its_two_neighbours():
    pv = small_pv()
    F = pv.shape[1]
    for pos in (0.0, 3.0, 3.25, 7.5, F - 1.0, F - 1.5):
        got = O.get_frame(pv, pos)
        lo, hi = int(np.floor(pos)), int(np.ceil(pos))
        mix = np.float32(np.float32(pos) - np.float32(lo))
        ref = (np.float32(1) - mix) * pv[:, lo] + mix * pv[:, hi]
        assert np.array_equal(got[:, 0], ref.astype(np.float32)), pos
    assert np.array_equal(O.get_frame(pv, 4.0)[:, 0], pv[:, 4])
    assert np.array_equal(O.get_frame(pv, 4.3, interp=3)[:, 0], pv[:, 4])          # floor interpolator
    assert np.array_equal(O.get_frame(pv, 4.3, interp=4)[:, 0], pv[:, 5])          # ceil interpolator


def test_freeze_plan_follows_the_reference_loops():
    F = 40
    # one freeze of 5 frames at frame 10: frame 10 appears 5 times instead of once; one trailing frame stays empty
    t10 = 10 * HOP / SR
    src = O.freeze_plan(F, SR, HOP, [t10], [5 * HOP / SR])
    assert len(src) == F + 5
    assert list(src[:10]) == list(range(10)) and list(src[10:15]) == [10] * 5 and list(src[15:44]) == list(range(11, 40)) and src[44] == -1
    # a freeze of length 0 drops its frame
    src = O.freeze_plan(F, SR, HOP, [t10], [0.0])
    assert len(src) == F and list(src[:39]) == [i for i in range(40) if i != 10] and src[39] == -1
    # unsorted events, two on one frame (the first given survives), times clamped into the PV, negative lengths -> 0
    src = O.freeze_plan(F, SR, HOP, [30 * HOP / SR, t10, t10, 1e9, -5.0], [2 * HOP / SR, 3 * HOP / SR, 7 * HOP / SR, 1 * HOP / SR, -1.0])
    events = {0: 0, 10: 3, 30: 2, 39: 1}
    expect = []
    for i in range(F):
        expect += [i] * events[i] if i in events else [i]
    assert len(src) == F + 6 and list(src[:len(expect)]) == expect and np.all(src[len(expect):] == -1)
    # no events: the identity
    assert list(O.freeze_plan(F, SR, HOP, [], [])) == list(range(F))


def test_library_planning_helpers_match_the_checker():
    """flanhip_freeze_plan / flanhip_cut_frames_range are host arithmetic: they run without a device"""
    import ctypes as C
    import flan_amd
    rng = np.random.default_rng(3)
    for trial in range(200):
        F = int(rng.integers(1, 300))
        n = int(rng.integers(0, 9))
        times = rng.uniform(-0.5, F * HOP / SR * 1.2, n).astype(np.float32)
        if n > 2:
            times[1] = times[0]
        lengths = rng.uniform(-0.01, 0.05, n).astype(np.float32)
        assert np.array_equal(flan_amd.freeze_plan(F, SR, HOP, times, lengths), O.freeze_plan(F, SR, HOP, times, lengths)), trial
        start, end = int(rng.integers(-20, F + 20)), int(rng.integers(-20, F + 20))
        s1, c1, s2, c2 = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        flan_amd.check(flan_amd.lib.flanhip_cut_frames_range(F, start, end, C.byref(s1), C.byref(c1)))
        O.lib.oracle_cut_frames_range(F, start, end, C.byref(s2), C.byref(c2))
        assert (s1.value, c1.value) == (s2.value, c2.value)
        assert c1.value == (0 if end <= start else max(min(max(end, 0), F - 1) - min(max(start, 0), F - 1), 0))


def test_cut_and_join():
    pv = small_pv()
    F = pv.shape[1]
    assert O.cut_frames(pv, 5, 5) is None and O.cut_frames(pv, 9, 2) is None
    assert np.array_equal(O.cut_frames(pv, 3, 11), pv[:, 3:11])
    assert np.array_equal(O.cut_frames(pv, -4, F + 10), pv[:, 0:F - 1])             # the end clamps to F-1 (PV.cpp:653): the last frame is lost
    a, b = pv[:, :7], pv[:, 7:20]
    assert np.array_equal(O.join([a, b]), pv[:, :20])
    other = O.analyze(O.noise(3, 3000, seed=9), SR, 1024, HOP, 1024)                # more channels, more bins than the first input
    j = O.join([a, other])
    assert j.shape == (2, 7 + other.shape[1], pv.shape[2], 2)
    assert np.array_equal(j[:, 7:], other[:2, :, :pv.shape[2]])
    fewer = pv[:1, :4, :100]
    j = O.join([a, fewer])
    assert np.array_equal(j[0, 7:, :100], fewer[0]) and not j[1, 7:].any() and not j[0, 7:, 100:].any()


def test_select_against_numpy():
    pv = small_pv()
    ch, F, bins, _ = pv.shape
    dft = (bins - 1) * 2
    rng = np.random.default_rng(8)
    Fo = 25
    sel = np.empty((Fo, bins, 2), np.float32)
    sel[..., 0] = rng.uniform(-0.02, F * HOP / SR * 1.1, (Fo, bins))
    sel[..., 1] = rng.uniform(-200.0, SR / 2 * 1.1, (Fo, bins))
    sel[3, 7] = (np.nan, 100.0)
    sel[3, 8] = (0.01, np.nan)
    sel[4, 9] = (0.01, 0.5)                                                         # s.f <= 1: the frequency is not rescaled
    got = O.select(pv, SR, HOP, sel)
    ref = np.zeros_like(got)
    with np.errstate(invalid="ignore"):
        sf_f = (sel[..., 0] * np.float32(SR) / np.float32(HOP)).astype(np.float32)
        sb_f = (sel[..., 1] / (np.float32(SR) / np.float32(dft))).astype(np.float32)
    for fr in range(Fo):
        for b in range(bins):
            if np.isnan(sf_f[fr, b]) or np.isnan(sb_f[fr, b]):
                continue
            sf, sb = int(sf_f[fr, b]), int(sb_f[fr, b])
            if sf < 0 or F - 1 <= sf or sb < 0 or bins - 1 <= sb:
                continue
            m = pv[:, sf, sb].copy()
            if sel[fr, b, 1] > 1:
                m[:, 1] = m[:, 1] * np.float32(np.float32(np.float32(b) * np.float32(SR) / np.float32(dft)) / sel[fr, b, 1])
            ref[:, fr, b] = m
    assert np.array_equal(got, ref)
    assert got.any()


def test_harmonic_scale_against_numpy():
    pv = small_pv(ch=1, n=4000, dft=256)
    ch, F, bins, _ = pv.shape
    dft = (bins - 1) * 2
    rng = np.random.default_rng(2)
    for mode, H in ((0, 15), (1, bins)):
        series = rng.uniform(-0.2, 1.0, (F, H)).astype(np.float32)
        got = O.harmonic_scale(pv, SR, series, mode)
        ref = np.zeros_like(pv)
        for fr in range(F):
            for b in range(bins):
                m, f = pv[0, fr, b]
                if f <= 1.0:
                    continue
                for h in range(H):
                    hf = np.float32(np.float64(f) * 2.0 ** (h + 1)) if mode == 0 else np.float32(f * np.float32(h + 2))
                    hb = int(np.float32(hf / (np.float32(SR) / np.float32(dft))))
                    if hb >= bins:
                        break
                    mag = np.float32(m * series[fr, h])
                    if ref[0, fr, hb, 0] < mag:
                        ref[0, fr, hb] = (mag, hf)
        assert np.array_equal(got, ref), mode
        assert got.any()


def test_smear_time_against_numpy_and_plan_helpers_agree():
    import flan_amd
    pv = small_pv(ch=1, n=5000, dft=128)
    ch, F, bins, _ = pv.shape
    rng = np.random.default_rng(12)
    ar = np.float32(SR) / np.float32(HOP)
    for smear, gran in ((0.02, 1), (0.05, 3), (rng.uniform(-0.01, 0.06, (F, bins)).astype(np.float32), rng.integers(-1, 5, (F, bins)).astype(np.int32)), (0.0, 5)):
        plan = O.smear_time_plan(F, bins, SR, HOP, smear)
        assert plan == flan_amd.smear_time_plan(F, bins, SR, HOP, smear)              # host arithmetic of the library, no device
        left, Fo, half = plan
        dist = O.smear_distribution(half)
        assert len(dist) == 2 * half
        got = O.smear_time(pv, SR, HOP, smear, gran, dist, left, Fo)
        ref = np.zeros((1, Fo, bins, 2), np.float32)
        for of in range(Fo):
            inf = min(max(of + left, 0), F - 1)
            for b in range(bins):
                sz = np.float32(max(smear if np.isscalar(smear) else smear[inf, b], 0.0))
                e = int(np.float32(sz * np.float32(SR) / np.float32(HOP)))
                g = max(int(gran if np.isscalar(gran) else gran[inf, b]), 1)
                ms = fs = tw = uw = 0.0
                for off in range(-e, e, g):
                    di = np.float32(np.float32(np.float32(off) / ar) / sz)
                    acc = int(np.float32(np.float32(np.float32(len(dist)) * np.float32(0.5)) * np.float32(np.float32(1) + di)))
                    acc = min(max(acc, 0), len(dist) - 1)
                    d = dist[acc]
                    tw += float(d)
                    src = of + left + off
                    if src < 0 or src >= F:
                        continue
                    uw += float(d)
                    ms += float(np.float32(pv[0, src, b, 0] * d))
                    fs += float(np.float32(pv[0, src, b, 1] * d))
                if tw > 0:
                    ms /= tw
                if uw > 0:
                    fs /= uw
                ref[0, of, b] = (np.float32(ms), np.float32(fs))
        assert np.array_equal(got, ref)
    # no smear at all: the output is the input minus its last frame (PVModify.cpp:563: rightmost - leftmost), every MF zero
    # because a point that spreads over no frames averages nothing (:580)
    left, Fo, half = O.smear_time_plan(F, bins, SR, HOP, 0.0)
    assert (left, Fo, half) == (0, F - 1, 0)


def exact_grid(F, bins, dft, hop):
    """the input's own ( time, frequency ) grid, with rates that make frame <-> time exact in fp32 (48000 / 375 = 128)"""
    t = (np.arange(F, dtype=np.float32) / np.float32(SR / hop))[:, None]
    f = (np.arange(bins, dtype=np.float32) * np.float32(SR) / np.float32(dft))[None, :]
    return np.stack(np.broadcast_arrays(t, f), axis=-1).astype(np.float32)


def test_modify_on_maps_with_known_answers():
    """PV::modify (PVModify.cpp:15-193) where the rasterisation has a closed form"""
    hop, dft = 375, 128
    pv = O.analyze(O.noise(2, 4000, seed=5), SR, dft, hop, dft)
    ch, F, bins, _ = pv.shape
    ident = exact_grid(F, bins, dft, hop)
    in_f = np.random.default_rng(1).uniform(0, 20000, (ch, F, bins)).astype(np.float32)
    expect = np.stack([pv[..., 0], in_f], axis=-1)
    # identity: every grid point is the ( l, m ) = ( 0, 0 ) corner of exactly one quad; the last frame and bin belong to none
    assert O.modify_out_frames(ident, SR, hop) == F - 1
    out = O.modify(pv, SR, hop, ident, in_f)
    assert out.shape == (ch, F - 1, bins, 2)
    assert np.array_equal(out[:, :, :bins - 1], expect[:, :F - 1, :bins - 1]) and not out[:, :, bins - 1].any()
    # a shift by three whole frames
    sh = ident.copy(); sh[..., 0] += np.float32(3 * hop / SR)
    out = O.modify(pv, SR, hop, sh, in_f)
    assert out.shape[1] == F + 2 and np.array_equal(out[:, 3:, :bins - 1], expect[:, :F - 1, :bins - 1]) and not out[:, :3].any()
    # twice as slow: the frames in between take the louder of their two half-weighted neighbours, with that neighbour's frequency
    st = ident.copy(); st[..., 0] *= 2
    out = O.modify(pv, SR, hop, st, in_f)
    assert np.array_equal(out[:, 0:2 * (F - 1):2, :bins - 1], expect[:, :F - 1, :bins - 1])
    a, b = np.float32(0.5) * pv[:, :F - 1, :bins - 1, 0], np.float32(0.5) * pv[:, 1:F, :bins - 1, 0]
    assert np.array_equal(out[:, 1:2 * (F - 1):2, :bins - 1, 0], np.maximum(a, b))
    assert np.array_equal(out[:, 1:2 * (F - 1):2, :bins - 1, 1], np.where(a < b, in_f[:, 1:F, :bins - 1], in_f[:, :F - 1, :bins - 1]))
    # the interpolators that ignore their argument: midpoint weighs every corner by a quarter
    out = O.modify(pv, SR, hop, ident, in_f, interp=1)
    corners = np.stack([pv[:, :F - 1, :bins - 1, 0], pv[:, 1:, :bins - 1, 0], pv[:, 1:, 1:, 0], pv[:, :F - 1, 1:, 0]])
    assert np.array_equal(out[:, :, :bins - 1, 0], (np.float32(0.25) * corners).max(axis=0))
    # longer than ten minutes: refused (:30-34); everything mapped before time zero: nothing to make
    far = ident.copy(); far[..., 0] += np.float32(601.0)
    assert O.modify_out_frames(far, SR, hop) == -2
    back = ident.copy(); back[..., 0] -= np.float32(100.0)
    assert O.modify_out_frames(back, SR, hop) == 0
    import flan_amd
    for g in (ident, sh, st, far, back):
        assert flan_amd.modify_out_frames(g, SR, hop) == O.modify_out_frames(g, SR, hop)


def test_stretch_spline_properties_and_plan_helper():
    """the checker's stretch_spline (pinned against the real spline.h in test_oracle_vs_ref.py) on data with known splines"""
    import flan_amd
    F, bins = 40, 5
    pv = np.zeros((1, F, bins, 2), np.float32)
    k = np.arange(F, dtype=np.float32)
    pv[0, :, 0, 0] = 3.0 + 0.5 * k                                                    # a straight line over uniform knots is its own natural spline
    pv[0, :, 1, 0] = 7.0
    pv[0, :, 2, 1] = 1000.0 - 2.0 * k
    steps = np.full(F - 1, 4, np.uint32)
    out = O.stretch_spline(pv, steps)
    assert out.shape == (1, 4 * (F - 1), bins, 2)
    t = np.arange(4 * (F - 1), dtype=np.float64) / 4.0
    assert np.allclose(out[0, :, 0, 0], 3.0 + 0.5 * t, rtol=0, atol=1e-5)
    assert np.array_equal(out[0, :, 1, 0], np.full(len(t), 7.0, np.float32))
    assert np.allclose(out[0, :, 2, 1], 1000.0 - 2.0 * t, rtol=0, atol=1e-3)
    assert not out[0, :, 3:].any()
    # the library's step bookkeeping (host arithmetic) against the checker's
    rng = np.random.default_rng(2)
    for trial in range(50):
        n = int(rng.integers(3, 200))
        st = rng.integers(1, 1000, n - 1).astype(np.uint32)
        assert flan_amd.lib.flanhip_stretch_spline_out_frames(st.ctypes.data, n) == O.lib.oracle_stretch_spline_out_frames(st.ctypes.data, n) == int(st.sum())
    assert flan_amd.lib.flanhip_stretch_spline_out_frames(np.ones(1, np.uint32).ctypes.data, 2) == -1
    assert flan_amd.lib.flanhip_stretch_spline_out_frames(np.array([1, 0, 1], np.uint32).ctypes.data, 4) == -1


def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "processors", "processors_arrange.npz"))


def golden_outputs(impl, g):
    """every method of the fixture through `impl` (the checker here, the GPU library in test_gpu_processors_arrange.py)"""
    pv, hop = g["pv"], int(g["hop"])
    left, Fs, half = (int(v) for v in g["smear_plan"])
    return dict(get_frame=impl.get_frame(pv, 7.25, 0), freeze=impl.freeze(pv, SR, hop, g["times"], g["lengths"]), cut=impl.cut_frames(pv, 3, 17),
                join=impl.join([pv[:, :5], pv[:, 9:]]), select=impl.select(pv, SR, hop, g["sel"]), octaves=impl.harmonic_scale(pv, SR, g["series_oct"], 0),
                harmonics=impl.harmonic_scale(pv, SR, g["series_har"], 1), smear_time=impl.smear_time(pv, SR, hop, g["smear"], 2, g["dist"], left, Fs),
                modify=impl.modify(pv, SR, hop, g["warp"], g["in_f"], 0, int(g["modify_frames"])), stretch_spline=impl.stretch_spline(pv, g["steps"]))


def test_checker_reproduces_the_golden_fixture():
    """tests/golden/processors/processors_arrange.npz (written by tests/golden/make_golden.py): the checker must not drift"""
    g = _golden()
    assert (int(g["smear_plan"][0]), int(g["smear_plan"][1]), int(g["smear_plan"][2])) == O.smear_time_plan(g["pv"].shape[1], g["pv"].shape[2], SR, int(g["hop"]), g["smear"])
    assert int(g["modify_frames"]) == O.modify_out_frames(g["warp"], SR, int(g["hop"]))
    for name, got in golden_outputs(O, g).items():
        assert got.shape == g[name].shape and np.array_equal(got.view(np.uint32), g[name].view(np.uint32)), name
