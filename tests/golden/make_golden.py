#!/usr/bin/env python3
"""Regenerate the golden vectors under tests/golden/ from the CPU oracle (oracle/flan_oracle.cpp).

The reference ships no fixtures for this path and its frame loops cannot be built here (FFTW3f / libsndfile absent), so
these vectors are the oracle's own outputs: they pin the oracle against regressions and give the GPU tests fixed
expected values that travel to the GPU box.  The oracle itself is pinned against the reference's buildable translation
units (tests/test_oracle_vs_ref.py) and the SURVEY 8c anchors (tests/test_oracle_anchors.py).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

SR = 48000.0
CASES = {
    # name: (channels, n, window, hop, dft, kind, seed)
    "sine_0p25s": (1, 12000, 2048, 512, 2048, "sine", 0),
    "noise_stereo_0p2s": (2, 9600, 2048, 512, 2048, "noise", 1234),
    "noise_dft4096_hop128": (1, 6000, 2048, 128, 4096, "noise", 7),
    "ragged_len": (1, 5003, 1024, 256, 1024, "noise", 11),
    "one_frame": (1, 100, 2048, 512, 2048, "noise", 13),
    "all_zero": (1, 3000, 1024, 256, 1024, "zeros", 0),
}


def make_input(kind, ch, n, seed):
    if kind == "sine":
        return O.sine(n)
    if kind == "zeros":
        return np.zeros((ch, n), np.float32)
    return O.noise(ch, n, seed)


def ext_cases(pv):
    """inputs + oracle outputs of the further frame processors (oracle/processors_oracle.cpp) on one small PV"""
    ch, F, bins, _ = pv.shape
    rng = np.random.default_rng(99)
    amount = rng.uniform(-0.2, 1.2, (F, bins)).astype(np.float32)
    src = pv[:, ::-1].copy()
    n = rng.integers(0, 60, F).astype(np.int32)
    ratio = rng.uniform(0.0, 0.6, (F, bins)).astype(np.float32)
    start, end, Fo = 2, 12, 30
    samples = O.time_extrapolate_interp_samples(start, end, Fo, 0)
    return dict(pv=pv, amount=amount, src=src, n=n, ratio=ratio, te_params=np.array([start, end, Fo], np.int64), te_samples=samples,
                replace=O.replace_amplitudes(pv, src, amount), subtract=O.subtract_amplitudes(pv, src, amount),
                resonate=O.resonate(pv, SR, 256, 0.05, 0.5, pow_mode=0), retain=O.n_loudest_partials(pv, n, False),
                remove=O.n_loudest_partials(pv, n, True), desample=O.desample(pv, ratio, 0),
                time_extrapolate=O.time_extrapolate(pv, SR, start, end, Fo, samples))


def arrange_cases(pv):
    """inputs + oracle outputs of the frame-selecting / warping methods (oracle/arrange_oracle.cpp) on one small PV"""
    ch, F, bins, _ = pv.shape
    hop, dft = 256, (bins - 1) * 2
    rng = np.random.default_rng(123)
    t = (np.arange(F, dtype=np.float32) / np.float32(SR / hop))[:, None] * np.ones((1, bins), np.float32)
    f = (np.arange(bins, dtype=np.float32) * np.float32(SR) / np.float32(dft))[None, :] * np.ones((F, 1), np.float32)
    times = np.array([0.02, 0.05, 0.05], np.float32)
    lengths = np.array([0.01, 0.02, 0.5], np.float32)
    sel = np.stack([rng.uniform(-0.01, F * hop / SR, (30, bins)), rng.uniform(-100, SR / 2, (30, bins))], -1).astype(np.float32)
    series_oct = rng.uniform(0, 1, (F, 12)).astype(np.float32)
    series_har = rng.uniform(0, 1, (F, 40)).astype(np.float32)
    smear = rng.uniform(0, 0.02, (F, bins)).astype(np.float32)
    left, Fs, half = O.smear_time_plan(F, bins, SR, hop, smear)
    dist = O.smear_distribution(half)
    warp = np.stack([t * np.float32(1.5) + np.float32(0.003) * np.sin(f / np.float32(2500)).astype(np.float32), f * np.float32(1.2) + np.float32(30)], -1).astype(np.float32)
    in_f = rng.uniform(0, SR / 2, (ch, F, bins)).astype(np.float32)
    Fm = O.modify_out_frames(warp, SR, hop)
    steps = rng.integers(1, 5, F - 1).astype(np.uint32)
    return dict(pv=pv, hop=np.int64(hop), times=times, lengths=lengths, sel=sel, series_oct=series_oct, series_har=series_har, smear=smear,
                smear_plan=np.array([left, Fs, half], np.int64), dist=dist, warp=warp, in_f=in_f, modify_frames=np.int64(Fm), steps=steps,
                get_frame=O.get_frame(pv, 7.25, 0), freeze=O.freeze(pv, SR, hop, times, lengths), cut=O.cut_frames(pv, 3, 17),
                join=O.join([pv[:, :5], pv[:, 9:]]), select=O.select(pv, SR, hop, sel), octaves=O.harmonic_scale(pv, SR, series_oct, 0),
                harmonics=O.harmonic_scale(pv, SR, series_har, 1), smear_time=O.smear_time(pv, SR, hop, smear, 2, dist, left, Fs),
                modify=O.modify(pv, SR, hop, warp, in_f, 0, Fm), stretch_spline=O.stretch_spline(pv, steps))


def main():
    pv = O.analyze(O.noise(1, 5003, 11), SR, 1024, 256, 1024)
    os.makedirs(os.path.join(HERE, "processors"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "processors", "processors_ext.npz"), **ext_cases(pv))
    # 96 -> 48 kHz of a 2-channel 0.1 s buffer (pins the cross-channel bleed: r8brain sees one stream, AudioConversions.cpp:14-30),
    # and one up-sampling ratio; the resampler restated in oracle/resample_oracle.cpp is itself pinned against the vendored r8brain
    x96 = O.noise(2, 9600, 21)
    x32 = O.noise(2, 3200, 22)
    np.savez_compressed(os.path.join(HERE, "processors", "resample.npz"), x96=x96, y48=O.resample_2to1(x96, 96000.0, 48000.0),
                        x32=x32, y48_from_32=O.resample_rational(x32, 32000.0, 48000.0, 3, 2))
    small = O.analyze(O.noise(2, 4000, 17), SR, 256, 256, 256)
    np.savez_compressed(os.path.join(HERE, "processors", "processors_arrange.npz"), **arrange_cases(small))
    for name, (ch, n, W, hop, dft, kind, seed) in CASES.items():
        x = make_input(kind, ch, n, seed)
        pv = O.analyze(x, SR, W, hop, dft)
        ar = np.float32(SR) / np.float32(hop)
        out, _ = O.synthesize(pv, SR, ar, W)
        extra = {}
        if name == "noise_stereo_0p2s":
            F, bins = pv.shape[1], pv.shape[2]
            two = np.full((F, bins), 2.0, np.float32)
            extra["stretch2"] = O.stretch(pv, SR, hop, two)
            extra["repitch2"] = O.repitch(pv, SR, two)
            extra["shape_f_plus_100"] = O.shape_affine(pv, SR, 1.0, 0.0, 1.0, 100.0, False)
            extra["shape_f_times_2_aligned"] = O.shape_affine(pv, SR, 1.0, 0.0, 2.0, 0.0, True)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), audio=x, pv=pv, out=out,
                            params=np.array([ch, n, W, hop, dft], np.int64), **extra)
        print(name, pv.shape, out.shape)


if __name__ == "__main__":
    main()
