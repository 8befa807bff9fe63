"""GPU parity of Audio::resample (2:1, the r8brain path of BASELINE config 5) through the C ABI, and config 5 in small."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import flan_amd
    assert flan_amd.lib.flanhip_device_count() > 0
    return flan_amd


@pytest.mark.parametrize("ch,n", [(2, 19200), (1, 20001), (2, 4801), (3, 1000), (1, 100), (2, 600000)])
def test_resample_2to1_parity(fa, ch, n):
    x = O.noise(ch, n, seed=n)
    ref = O.resample_2to1(x, 96000.0, 48000.0)
    got = fa.resample(x, 96000.0, 48000.0)
    assert got.shape == ref.shape
    d = got.astype(np.float64) - ref.astype(np.float64)
    same = np.mean(got.view(np.uint32) == ref.view(np.uint32))
    print("\n[resample %dx%d] rms diff %.2e  max %.2e  bit-identical %.5f" % (ch, n, np.sqrt(np.mean(d ** 2)), np.abs(d).max(), same))
    assert np.sqrt(np.mean(d ** 2)) <= 1e-5          # north-star tolerance; measured ~1e-11 (fma vs mul+add in the 16th digit)
    assert np.abs(d).max() <= 2e-7


@pytest.mark.parametrize("src,dst,up,down", [(144000.0, 48000.0, 1, 3), (72000.0, 48000.0, 2, 3), (32000.0, 48000.0, 3, 2), (64000.0, 48000.0, 3, 4),
                                             (48000.0, 96000.0, 2, 1), (16000.0, 48000.0, 3, 1), (88200.0, 44100.0, 1, 2)])
def test_resample_single_step_ratios(fa, src, dst, up, down):
    """the other ratios r8brain serves with one block convolver (k_resample_up / k_resample_down / k_resample_rational), ragged lengths, several channels"""
    for ch, n in ((2, 30001), (3, 777), (1, 50)):
        x = O.noise(ch, n, seed=n + up)
        ref = O.resample_rational(x, src, dst, up, down)
        got = fa.resample(x, src, dst)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        same = np.mean(got.view(np.uint32) == ref.view(np.uint32))
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max() if d.size else 0.0, same if d.size else 1.0))
        assert (d.max() if d.size else 0.0) <= 2e-7 and (same >= 0.999 if d.size else True)


TWO_STAGE = [(44100.0, 48000.0), (48000.0, 44100.0), (96000.0, 44100.0), (22050.0, 48000.0), (44100.0, 96000.0), (44100.0, 32000.0),
             (32000.0, 44100.0), (44100.0, 12000.0), (11025.0, 8000.0)]


@pytest.mark.parametrize("src,dst", TWO_STAGE)
def test_resample_two_stage_ratios(fa, src, dst):
    """44.1 <-> 48 kHz and the other block convolver + whole-stepping interpolator ratios (k_resample_up + k_frac_whole)
    against the restatement, itself bit-identical to the real r8brain on these rates (test_oracle_resample.py): ragged lengths,
    several channels (one stream: the ringing crosses channel boundaries), inputs shorter than either filter"""
    assert O.two_stage_shape(src, dst) is not None
    for ch, n in ((2, 30001), (3, 777), (1, 50)):
        x = O.noise(ch, n, seed=n + int(dst))
        ref = O.resample_two_stage(x, src, dst)
        got = fa.resample(x, src, dst)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        same = np.mean(got.view(np.uint32) == ref.view(np.uint32))
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max() if d.size else 0.0, same if d.size else 1.0))
        assert (d.max() if d.size else 0.0) <= 1.2e-7 and (same >= 0.999 if d.size else True)


HB_CHAINS = [(48000.0, 192000.0), (48000.0, 384000.0), (16000.0, 96000.0), (8000.0, 96000.0), (6000.0, 96000.0), (192000.0, 48000.0), (96000.0, 16000.0),
             (192000.0, 24000.0), (192000.0, 44100.0), (384000.0, 16000.0), (768000.0, 48000.0)]


@pytest.mark.parametrize("src,dst", HB_CHAINS)
def test_resample_half_band_chains(fa, src, dst):
    """r8brain's chains with half-band stages -- 4x / 8x / 16x / 6x / 12x up (block convolver + CDSPHBUpsamplers), src >= 4 dst down
    (CDSPHBDownsamplers + block convolver [+ whole-stepping interpolator]) -- against the restatement, which test_oracle_resample.py holds
    to the real r8brain on the same rate pairs"""
    assert O.chain_shape(src, dst) is not None
    for ch, n in ((2, 20001), (3, 777), (1, 40)):
        x = O.noise(ch, n, seed=n + int(dst))
        ref = O.resample_chain(x, src, dst)
        got = fa.resample(x, src, dst)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        same = np.mean(got.view(np.uint32) == ref.view(np.uint32)) if d.size else 1.0
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max() if d.size else 0.0, same))
        assert (d.max() if d.size else 0.0) <= 1.2e-7 and same >= 0.999


SPLINE = [(44100.0, 48001.0), (48000.0, 50854.3), (44100.0, 22000.0), (44100.0, 44056.0), (96000.0, 44101.0), (44100.0, 14000.3)]


@pytest.mark.parametrize("src,dst", SPLINE)
def test_resample_without_whole_stepping(fa, src, dst):
    """rates with no small common divisor: r8brain's interpolator then reads a spline-interpolated bank of 1893 (2329 third-band) fractional
    delay filters at fp64 positions whose counter is re-based at every process() call (k_frac_spline; the call boundaries are worked out on
    the host: spline_segments).  Several channels = several calls = several re-basings; against the restatement, which is bit-identical to
    the vendored r8brain on these rates (test_oracle_resample.py)"""
    src, dst = float(np.float32(src)), float(np.float32(dst))           # FrameRate is a float (Audio.h): r8brain sees the float's value
    sh = O.chain_shape(src, dst)
    assert sh is not None and sh.get("spline")
    for ch, n in ((4, 20001), (3, 2500), (1, 300)):
        x = O.noise(ch, n, seed=n + int(dst))
        ref = O.resample_chain(x, src, dst)
        got = fa.resample(x, src, dst)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        same = np.mean(got.view(np.uint32) == ref.view(np.uint32))
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max(), same))
        assert d.max() <= 1.2e-7 and same >= 0.999


GENERAL = [(8000.0, 44100.0), (11025.0, 48000.0), (8000.0, 48001.0), (44100.0, 192000.0), (192000.0, 44101.0), (96000.0, 11026.0), (1000.0, 44100.0)]


@pytest.mark.parametrize("src,dst", GENERAL)
def test_resample_general_chains(fa, src, dst):
    """the rest of CDSPResampler's constructor: upsampling with intermediate interpolation (2x convolver -> interpolator -> a 2x convolver whose
    transition band follows from the rates -> half-band upsamplers: 8 -> 44.1 kHz, 44.1 -> 192 kHz ...) and half-band downsamplers in front of
    the spline-interpolated bank; the stage-list form of the checker is bit-identical to the vendored r8brain on these (test_oracle_resample.py)"""
    assert O.resample_stages(src, dst) is not None
    for ch, n in ((3, 5001), (2, 777), (1, 40)):
        x = O.noise(ch, n, seed=n + int(dst))
        ref = O.resample_general(x, src, dst)
        got = fa.resample(x, src, dst)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        same = np.mean(got.view(np.uint32) == ref.view(np.uint32)) if d.size else 1.0
        print("\n[resample %g->%g %dx%d] %s max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, O.resample_stages(src, dst), d.max() if d.size else 0.0, same))
        assert (d.max() if d.size else 0.0) <= 1.2e-7 and same >= 0.999


def test_resample_two_stage_long(fa):
    """a minute of 44.1 kHz stereo to 48 kHz: the sine comes out a sine (size-independent property; the oracle is not run at this size)"""
    sr, n = 44100.0, 44100 * 60
    t = np.arange(n, dtype=np.float64)
    x = np.stack([0.5 * np.sin(2 * np.pi * 1000.0 * t / sr), 0.25 * np.sin(2 * np.pi * 15000.0 * t / sr)]).astype(np.float32)
    y = fa.resample(x, sr, 48000.0)
    assert y.shape == (2, int(O.lib.oracle_resample_out_frames(n, sr, 48000.0)))
    # Audio::resample scales the frame count by the float ratio and walks the buffer as ONE stream: channel 1 starts where channel 0's
    # n*48000/44100 samples end, a fraction of a sample away from y.shape[1]
    u = np.arange(y.shape[1], dtype=np.float64)
    mid = slice(4000, y.shape[1] - 4000)
    assert np.abs(y[0, mid] - 0.5 * np.sin(2 * np.pi * 1000.0 * u[mid] / 48000.0)).max() <= 2e-6
    exact = n * 48000.0 / 44100.0
    shift = y.shape[1] - exact                        # channel 1 is read `shift` output samples late
    assert np.abs(y[1, mid] - 0.25 * np.sin(2 * np.pi * 15000.0 * (u[mid] + shift) / 48000.0)).max() <= 2e-6


def test_extreme_ratios(fa):
    """nothing is refused any more: half-band chains of any depth (the kernels r8brain picks at SteepIndex 4, 5, 6+ too) -- 64x up, 96x and 512x
    down, 480x up through intermediate interpolation with a 26.7 % transition band -- against the stage-list checker (bit-identical to the
    vendored r8brain on these: test_oracle_resample.py)"""
    for src, dst, ch, n in ((1000.0, 64000.0, 2, 1501), (96000.0, 1000.0, 2, 30001), (768000.0, 1500.0, 1, 200001), (100.0, 48000.0, 1, 301)):
        x = O.noise(ch, n, seed=n)
        ref = O.resample_general(x, src, dst)
        got = fa.resample(x, src, dst)
        assert got.shape == ref.shape
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        same = np.mean(got.view(np.uint32) == ref.view(np.uint32))
        print("\n[resample %g->%g %dx%d] %s max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, O.resample_stages(src, dst), d.max(), same))
        assert d.max() <= 1.2e-7 and same >= 0.999
    import flan_amd
    with pytest.raises(flan_amd.FlanHipError):
        fa.resample(O.noise(1, 100, seed=1), 48000.0, 0.0)                      # not a rate


def test_config5_small(fa):
    """BASELINE config 5 in small: stereo 96 kHz noise -> resample(48000) -> convert_to_PV(2048,512,2048) -> shape(f + 100 Hz) ->
    convert_to_audio; every stage against the oracle on the oracle's own previous stage."""
    x96 = O.noise(2, 96000, seed=1234)
    sr = 48000.0
    x48_r = O.resample_2to1(x96, 96000.0, sr)
    x48_g = fa.resample(x96, 96000.0, sr)
    assert np.abs(x48_g.astype(np.float64) - x48_r).max() <= 2e-7
    pv_r = O.analyze(x48_r, sr, 2048, 512, 2048)
    pv_g = fa.analyze(x48_r, sr, 2048, 512, 2048)
    m_g, m_r = pv_g[..., 0].astype(np.float64), pv_r[..., 0].astype(np.float64)
    assert np.sqrt(np.sum((m_g - m_r) ** 2) / np.sum(m_r ** 2)) <= 1e-5
    sh_r = O.shape_affine(pv_r, sr, 1.0, 0.0, 1.0, 100.0, False)
    sh_g = fa.shape_affine(pv_r, sr, 1.0, 0.0, 1.0, 100.0, False)
    assert np.array_equal(sh_g.view(np.uint32), sh_r.view(np.uint32))
    out_r, _ = O.synthesize(sh_r, sr, sr / 512, 2048)
    out_g, _ = fa.synthesize(sh_r, sr, sr / 512, 2048)
    rms = float(np.sqrt(np.mean((out_g.astype(np.float64) - out_r.astype(np.float64)) ** 2)))
    print("\n[config5-small] P2 rms %.3e" % rms)
    assert rms <= 1e-5


def test_resample_fixture(fa):
    """the committed vectors of tests/golden/processors/resample.npz through the GPU library: bit for bit"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "processors", "resample.npz"))
    assert np.array_equal(fa.resample(g["x96"], 96000.0, 48000.0).view(np.uint32), g["y48"].view(np.uint32))
    assert np.array_equal(fa.resample(g["x32"], 32000.0, 48000.0).view(np.uint32), g["y48_from_32"].view(np.uint32))


@pytest.mark.parametrize("tag", ["c5_stereo_0p1s", "c5_mono_ragged", "c5_stereo_0p25s", "c5_three_short", "up_32_48", "down_144_48", "down_72_48",
                                 "up_48_96", "up_16_48", "down_64_48", "ms_441_48", "ms_48_441", "hb_48_192", "hb_192_48", "hb_192_441",
                                 "hb_8_96", "hb_96_16", "sp_441_48001", "sp_48_50854", "sp_441_14000"])
def test_resample_against_the_real_r8brain(fa, tag):
    """the HIP resampler against vectors the reference's vendored r8brain produced (tests/golden/ref_made/r8brain.npz, made by
    make_ref_made.py from oracle/_ref/libr8bref.so): config 5's 96 -> 48 kHz incl. the 2-channel cross-channel bleed, every other
    single-step ratio, and 44.1 <-> 48 kHz (block convolver + fractional interpolator).  >= 99.9 % of the samples bit-identical, the rest within one fp32 ulp at unit scale."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_made", "r8brain.npz"))
    x, y = g[tag + "_x"], g[tag + "_y"]
    src, dst = (float(v) for v in g[tag + "_rates"])
    got = fa.resample(x, src, dst)
    assert got.shape == y.shape
    same = np.mean(got.view(np.uint32) == y.view(np.uint32))
    worst = np.abs(got.astype(np.float64) - y.astype(np.float64)).max()
    print("\n[%s vs real r8brain] bit-identical %.5f  worst %.2e" % (tag, same, worst))
    assert same >= 0.999 and worst <= 1.2e-7


@pytest.mark.parametrize("ch,n", [(1, 19808), (1, 19810), (1, 39615), (2, 24760), (3, 9907), (1, 250000)])
def test_resample_fft_convolver_against_the_direct_sums(fa, ch, n):
    """the 2:1 block convolver as fp64 overlap-save FFT convolution (k_resample_ols3, what streams of 8 blocks and more take) against the
    direct fp64 sums in the checker's order (the resample_direct hook of flanhip_debug_option): stream lengths at the switch-over, at whole numbers of block
    pairs, one past them, ragged; >= 99.9 % of the samples bit-identical, the rest one fp32 ulp at unit scale."""
    x = O.noise(ch, n, seed=n + ch)
    try:
        fa.lib.flanhip_debug_option(fa.DEBUG_RESAMPLE_DIRECT, 1)
        direct = fa.resample(x, 96000.0, 48000.0)
    finally:
        fa.lib.flanhip_debug_option(fa.DEBUG_RESAMPLE_DIRECT, 0)
    got = fa.resample(x, 96000.0, 48000.0)
    assert got.shape == direct.shape
    same = np.mean(got.view(np.uint32) == direct.view(np.uint32))
    worst = np.abs(got.astype(np.float64) - direct.astype(np.float64)).max()
    print("\n[fft vs direct %dx%d] bit-identical %.5f  worst %.2e" % (ch, n, same, worst))
    assert same >= 0.999 and worst <= 1.2e-7
    # the 256-thread radix-16 generation of the convolver (k_resample_ols2, hook value 2) against the same sums
    try:
        fa.lib.flanhip_debug_option(fa.DEBUG_RESAMPLE_DIRECT, 2)
        old = fa.resample(x, 96000.0, 48000.0)
    finally:
        fa.lib.flanhip_debug_option(fa.DEBUG_RESAMPLE_DIRECT, 0)
    assert np.mean(old.view(np.uint32) == direct.view(np.uint32)) >= 0.999 and np.abs(old.astype(np.float64) - direct.astype(np.float64)).max() <= 1.2e-7
    assert np.array_equal(direct.view(np.uint32), O.resample_2to1(x, 96000.0, 48000.0).view(np.uint32))    # the direct kernel IS the checker's sum


def test_config5_full_size_against_the_real_r8brain(fa):
    """BASELINE config 5's resample at its FULL size (stereo 60 s, 96 -> 48 kHz) against what the reference's vendored r8brain made of the same
    noise (tests/golden/ref_made/r8brain_config5_full.npz: six 32768-sample windows -- head, middle and tail of both channels, so the start-up,
    the cross-channel bleed at the channel boundary and the zero-flushed end are all in -- made by make_ref_made.py --config5-full)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_made", "r8brain_config5_full.npz"))
    x = O.noise(int(g["shape_in"][0]), int(g["shape_in"][1]), int(g["seed"]))
    got = fa.resample(x, float(g["rates"][0]), float(g["rates"][1]))
    assert got.shape == tuple(int(v) for v in g["shape_out"])
    for i, (c, a) in enumerate(g["windows"]):
        ref = g["w%d" % i]
        cur = got[int(c), int(a):int(a) + ref.size]
        same = np.mean(cur.view(np.uint32) == ref.view(np.uint32))
        worst = np.abs(cur.astype(np.float64) - ref.astype(np.float64)).max()
        print("\n[config 5 full size, channel %d from %d] bit-identical %.5f  worst %.2e" % (c, a, same, worst))
        assert same >= 0.999 and worst <= 1.2e-7
