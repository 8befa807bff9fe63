"""Audio::resample (2:1): the CPU restatement (oracle/resample_oracle.cpp) against the REAL r8brain resampler vendored by the
reference, compiled unmodified into oracle/_ref/libr8bref.so and driven exactly like Audio/AudioConversions.cpp:25-27."""
import numpy as np
import pytest

import oracle_lib as O

r8b = O.load_r8b_ref()
pytestmark = pytest.mark.skipif(r8b is None, reason="oracle/_ref/libr8bref.so not built (no /root/reference)")


def ref_resample(x, src, dst):
    ch, n = x.shape
    n_out = int(O.lib.oracle_resample_out_frames(n, src, dst))
    out = np.zeros((ch, n_out), np.float32)
    r8b.ref_r8b_resample(np.ascontiguousarray(x).reshape(-1), ch * n, float(src), float(dst), n, out.reshape(-1), ch * n_out)
    return out


def test_filter_is_the_1621_tap_design():
    taps = np.zeros(4000)
    n = O.lib.oracle_r8b_default_lowpass_half(taps, 4000)
    assert n == 1621                          # flt_len=1621 latency=810
    h = taps[:n]
    assert np.allclose(h, h[::-1], rtol=0, atol=1e-18) and abs(h.sum() - 1.0) < 1e-12


@pytest.mark.parametrize("ch,n", [(2, 9600), (1, 20001), (2, 4801), (3, 1000), (1, 100)])
def test_restatement_matches_r8brain(ch, n):
    rng = np.random.default_rng(ch * 1000 + n)
    x = rng.uniform(-1, 1, (ch, n)).astype(np.float32)
    ours = O.resample_2to1(x, 96000.0, 48000.0)
    theirs = ref_resample(x, 96000.0, 48000.0)
    assert ours.shape == theirs.shape
    d = ours.astype(np.float64) - theirs.astype(np.float64)
    same = np.mean(ours.view(np.uint32) == theirs.view(np.uint32))
    print("\n[resample %dx%d] rms diff %.2e  max %.2e  bit-identical %.5f" % (ch, n, np.sqrt(np.mean(d ** 2)), np.abs(d).max(), same))
    assert np.abs(d).max() <= 2e-7            # one fp32 ulp at unit scale: the fp64 sums differ in the 16th digit only
    assert same >= 0.995


def test_impulse_and_channel_bleed():
    """an impulse at the end of channel 0 rings into the start of channel 1: the buffer is one stream"""
    x = np.zeros((2, 4000), np.float32)
    x[0, 3999] = 1.0
    ours = O.resample_2to1(x, 96000.0, 48000.0)
    theirs = ref_resample(x, 96000.0, 48000.0)
    assert np.abs(theirs[1, :50]).max() > 1e-3
    assert np.abs(ours.astype(np.float64) - theirs).max() <= 2e-7


RATES = [(144000.0, 48000.0, 1, 3), (72000.0, 48000.0, 2, 3), (32000.0, 48000.0, 3, 2), (64000.0, 48000.0, 3, 4), (48000.0, 96000.0, 2, 1),
         (16000.0, 48000.0, 3, 1), (96000.0, 48000.0, 1, 2), (44100.0, 88200.0, 2, 1), (88200.0, 44100.0, 1, 2)]


@pytest.mark.parametrize("src,dst,up,down", RATES)
def test_single_step_ratios_match_r8brain(src, dst, up, down):
    """every ratio CDSPResampler serves with one block convolver (CDSPResampler.h:139-207): zero-stuff by `up`, the default
    low-pass at cut-off 1/max(up,down) with gain `up`, latency consumed, every `down`-th sample"""
    rng = np.random.default_rng(int(src + dst))
    for ch, n in ((2, 7001), (1, 300)):
        x = rng.uniform(-1, 1, (ch, n)).astype(np.float32)
        ours = O.resample_rational(x, src, dst, up, down)
        theirs = ref_resample(x, src, dst)
        assert ours.shape == theirs.shape
        d = np.abs(ours.astype(np.float64) - theirs.astype(np.float64))
        same = np.mean(ours.view(np.uint32) == theirs.view(np.uint32))
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max(), same))
        assert d.max() <= 3e-7 and same >= 0.995


def test_rational_form_contains_the_2to1_restatement():
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (2, 5000)).astype(np.float32)
    a = O.resample_2to1(x, 96000.0, 48000.0)
    b = O.resample_rational(x, 96000.0, 48000.0, 1, 2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


TWO_STAGE = [(44100.0, 48000.0), (48000.0, 44100.0), (96000.0, 44100.0), (22050.0, 48000.0), (44100.0, 96000.0), (44100.0, 32000.0),
             (32000.0, 44100.0), (44100.0, 12000.0), (48000.0, 22050.0), (11025.0, 8000.0)]


@pytest.mark.parametrize("src,dst", TWO_STAGE)
def test_two_stage_ratios_match_r8brain(src, dst):
    """the ratios CDSPResampler serves with one block convolver and one whole-stepping CDSPFracInterpolator (CDSPResampler.h:214-316,
    :319-378 without half-band stages): 2x zero-stuffing low-pass (or a low-pass in place when src >= 2 dst), then the bank of OutStep
    fractional-delay filters (third-band parameters for 44100 -> 12000)"""
    assert O.two_stage_shape(src, dst) is not None
    rng = np.random.default_rng(int(src + dst))
    for ch, n in ((2, 7001), (1, 300), (3, 20)):
        x = rng.uniform(-1, 1, (ch, n)).astype(np.float32)
        ours = O.resample_two_stage(x, src, dst)
        theirs = ref_resample(x, src, dst)
        assert ours.shape == theirs.shape
        d = np.abs(ours.astype(np.float64) - theirs.astype(np.float64))
        same = np.mean(ours.view(np.uint32) == theirs.view(np.uint32))
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max(), same))
        assert d.max() <= 1.2e-7 and same >= 0.999


HB_CHAINS = [(48000.0, 192000.0), (44100.0, 176400.0), (48000.0, 384000.0), (16000.0, 96000.0), (8000.0, 96000.0), (6000.0, 96000.0), (192000.0, 48000.0),
             (96000.0, 16000.0), (192000.0, 24000.0), (192000.0, 44100.0), (384000.0, 48000.0), (384000.0, 16000.0), (768000.0, 48000.0)]


@pytest.mark.parametrize("src,dst", HB_CHAINS)
def test_half_band_chains_match_r8brain(src, dst):
    """chains with CDSPHBUpsampler / CDSPHBDownsampler stages (CDSPResampler.h:174-212, :319-378): the restatement against the vendored
    r8brain, incl. third-band kernels (6x, 12x), three stages (16x up, 8x down of a 3:1) and half-band + interpolator (192 -> 44.1 kHz)"""
    assert O.chain_shape(src, dst) is not None
    rng = np.random.default_rng(int(src + dst))
    for ch, n in ((2, 7001), (1, 300), (3, 20)):
        x = rng.uniform(-1, 1, (ch, n)).astype(np.float32)
        ours = O.resample_chain(x, src, dst)
        theirs = ref_resample(x, src, dst)
        assert ours.shape == theirs.shape
        d = np.abs(ours.astype(np.float64) - theirs.astype(np.float64))
        same = np.mean(ours.view(np.uint32) == theirs.view(np.uint32)) if d.size else 1.0
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max() if d.size else 0.0, same))
        assert (d.max() if d.size else 0.0) <= 1.2e-7 and same >= 0.995


SPLINE = [(44100.0, 48001.0), (48000.0, 50854.3), (44100.0, 22000.0), (44100.0, 44056.0), (96000.0, 44101.0), (44100.0, 30000.5), (44100.0, 14000.3)]


@pytest.mark.parametrize("src,dst", SPLINE)
def test_ratios_without_whole_stepping_match_r8brain(src, dst):
    """no small common divisor: CDSPFracInterpolator with the spline-interpolated bank (getFilterBank( -1, 3, 8, ... ): FilterFracs from the
    ROUNDED attenuation, convolve2, the position counter re-based per process() call -- one call per channel's worth of input in oneshot)"""
    sh = O.chain_shape(src, dst)
    assert sh is not None and sh.get("spline")
    rng = np.random.default_rng(int(src + dst))
    for ch, n in ((4, 20001), (1, 300), (3, 2500)):
        x = rng.uniform(-1, 1, (ch, n)).astype(np.float32)
        ours = O.resample_chain(x, src, dst)
        theirs = ref_resample(x, src, dst)
        assert ours.shape == theirs.shape
        d = np.abs(ours.astype(np.float64) - theirs.astype(np.float64))
        same = np.mean(ours.view(np.uint32) == theirs.view(np.uint32))
        print("\n[resample %g->%g %dx%d] max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, d.max(), same))
        assert d.max() <= 1.2e-7 and same >= 0.999


GENERAL = [(1000.0, 64000.0), (96000.0, 1000.0), (500.0, 44100.0), (768000.0, 1500.0), (100.0, 48000.0), (8000.0, 44100.0), (11025.0, 48000.0), (8000.0, 48001.0), (16000.0, 44100.0), (44100.0, 192000.0), (22050.0, 96000.0), (192000.0, 44101.0),
           (96000.0, 11026.0), (1000.0, 44100.0), (44100.0, 48000.0), (96000.0, 48000.0), (192000.0, 44100.0), (8000.0, 96000.0)]


@pytest.mark.parametrize("src,dst", GENERAL)
def test_stage_list_matches_r8brain(src, dst):
    """the whole of CDSPResampler's constructor as a stage list (build_stages): upsampling with intermediate interpolation, half-band stages in
    front of the spline bank, and every shape the other forms of the checker restate"""
    assert O.resample_stages(src, dst) is not None
    rng = np.random.default_rng(int(src + dst))
    for ch, n in ((3, 5001), (1, 300)):
        x = rng.uniform(-1, 1, (ch, n)).astype(np.float32)
        ours = O.resample_general(x, src, dst)
        theirs = ref_resample(x, src, dst)
        assert ours.shape == theirs.shape
        if ours.size == 0:
            continue
        d = np.abs(ours.astype(np.float64) - theirs.astype(np.float64))
        same = np.mean(ours.view(np.uint32) == theirs.view(np.uint32))
        print("\n[resample %g->%g %dx%d] %s max diff %.2e  bit-identical %.5f" % (src, dst, ch, n, O.resample_stages(src, dst), d.max(), same))
        assert d.max() <= 1.2e-7 and same >= 0.995


def test_stage_lists():
    assert O.resample_stages(8000.0, 44100.0).split() == ["conv:2/1@0.5,tb2,g2", "frac:640/441", "conv:2/1@0.5,tb15.6787,g2", "hbup:0"]
    assert O.resample_stages(44100.0, 48000.0).split() == ["conv:2/1@0.5,tb2,g2", "frac:147/80"]
    assert O.resample_stages(192000.0, 44101.0).split()[0] == "hbdown:0" and O.resample_stages(192000.0, 44101.0).split()[-1].startswith("spline:")
    assert O.resample_stages(1000.0, 64000.0).split() == ["conv:2/1@0.5,tb2,g2"] + ["hbup:%d" % i for i in range(5)]
    assert O.resample_stages(96000.0, 1000.0).split()[:5] == ["hbdown:%dt" % i for i in (4, 3, 2, 1, 0)] and O.resample_stages(48000.0, 48000.0) is None
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (2, 4000)).astype(np.float32)
    for src, dst in ((96000.0, 48000.0), (44100.0, 48000.0), (192000.0, 48000.0), (44100.0, 48001.0)):
        assert np.array_equal(O.resample_general(x, src, dst).view(np.uint32), O.resample_chain(x, src, dst).view(np.uint32)), (src, dst)


def test_chain_form_contains_the_other_restatements():
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (2, 5000)).astype(np.float32)
    assert np.array_equal(O.resample_chain(x, 96000.0, 48000.0).view(np.uint32), O.resample_2to1(x, 96000.0, 48000.0).view(np.uint32))
    assert np.array_equal(O.resample_chain(x, 32000.0, 48000.0).view(np.uint32), O.resample_rational(x, 32000.0, 48000.0, 3, 2).view(np.uint32))
    assert np.array_equal(O.resample_chain(x, 44100.0, 48000.0).view(np.uint32), O.resample_two_stage(x, 44100.0, 48000.0).view(np.uint32))
    for src, dst in ((8000.0, 44100.0), (192000.0, 44101.0), (48000.0, 48000.0)):
        assert O.chain_shape(src, dst) is None, (src, dst)       # (the older Chain form: intermediate interpolation and half-band + spline are the stage list's)


def test_two_stage_shapes():
    assert O.two_stage_shape(44100.0, 48000.0) == dict(up=2, norm_freq=0.5, third=False, in_step=147, out_step=80)
    assert O.two_stage_shape(48000.0, 44100.0) == dict(up=2, norm_freq=0.459375, third=False, in_step=320, out_step=147)
    assert O.two_stage_shape(44100.0, 12000.0)["third"] is True and O.two_stage_shape(96000.0, 44100.0)["up"] == 1
    for src, dst in ((96000.0, 48000.0), (32000.0, 48000.0), (48000.0, 192000.0), (8000.0, 44100.0), (96000.0, 16000.0), (44100.0, 22000.0),
                     (44100.0, 48001.0), (48000.0, 48000.0)):
        assert O.two_stage_shape(src, dst) is None, (src, dst)       # single step / half-band stages / no whole stepping (chain_shape serves it) / same rate
    bank = O.frac_bank(80)
    assert bank.shape == (80, 28) and np.allclose(bank.sum(1), 1.0, rtol=0, atol=1e-14)
    assert bank[0, 13] == 1.0 and np.abs(np.delete(bank[0], 13)).max() < 1e-15      # row 0 is the unit delay: y passes through
    assert O.frac_bank(40, third=True).shape == (40, 22)


def test_filter_lengths_of_the_other_cutoffs():
    taps = np.zeros(8000)
    assert O.lib.oracle_r8b_default_lowpass(1.0 / 3.0, 1.0, taps, 8000) == 2431     # fl2 = 1215
    assert O.lib.oracle_r8b_default_lowpass(0.25, 3.0, taps, 8000) == 3241          # fl2 = 1620, DC gain 3
    assert abs(taps[:3241].sum() - 3.0) < 1e-11


def test_checker_reproduces_the_resample_fixture():
    """tests/golden/processors/resample.npz (96 -> 48 kHz of a 2-channel 0.1 s buffer, plus 32 -> 48 kHz): the checker must not drift"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "processors", "resample.npz"))
    assert np.array_equal(O.resample_2to1(g["x96"], 96000.0, 48000.0).view(np.uint32), g["y48"].view(np.uint32))
    assert np.array_equal(O.resample_rational(g["x32"], 32000.0, 48000.0, 3, 2).view(np.uint32), g["y48_from_32"].view(np.uint32))
    # the bleed itself: the tail of channel 0's filter response lands in the first samples of channel 1
    solo = g["x96"].copy(); solo[1] = 0.0
    assert np.any(O.resample_2to1(solo, 96000.0, 48000.0)[1] != 0.0)
