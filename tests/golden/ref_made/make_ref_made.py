#!/usr/bin/env python3
"""Vectors made by the REFERENCE ITSELF -- the translation units of /root/reference that compile here unmodified
(oracle/_ref/libflanref.so: phase_vocoder.cpp, WindowFunctions.cpp, PV/PVBuffer.cpp + Utility/Bytes.cpp, Utility/Interpolator.cpp;
oracle/_ref/libr8bref.so: the vendored r8brain) -- so that the pinned part of the oracle travels as DATA to machines that have no
/root/reference and no prebuilt reference library.  Inputs are counter-based noise / seeded numpy draws stored next to the outputs.

    python tests/golden/ref_made/make_ref_made.py        (needs oracle/_ref, i.e. /root/reference; run in the build container)

Files (all small):
    r8brain.npz         Audio::resample's one oneshot<float,float> over the whole channel-major buffer (AudioConversions.cpp:25-27) for
                        config 5's 96 -> 48 kHz (incl. the 2-channel case that pins the cross-channel bleed), the other single-step
                        ratios, and 44.1 <-> 48 kHz (multi-stage: refused by the HIP path today, kept for when it is built)
    phase_vocoder.npz   phase_vocoder() / inverse_phase_vocoder() (phase_vocoder.cpp:5-61) on random spectra and phases
    hann.npz            Windows::hann over the window sizes the path uses
    pvbuffer.npz        PVBuffer unit conversions and a .flan file image written by the reference's own save()
    interpolators.npz   the named Interpolators on a dense grid
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle_lib as O  # noqa: E402


def r8b(ref, x, src, dst):
    flat = np.ascontiguousarray(x, np.float32).reshape(-1)
    ch, n = x.shape
    n_out = int(n * (dst / src))                                # AudioConversions.cpp:22 (float product truncated)
    out = np.zeros(ch * n_out, np.float32)
    ref.ref_r8b_resample(flat, flat.size, float(src), float(dst), n, out, out.size)
    return out.reshape(ch, n_out)


def main():
    ref, r8 = O.load_ref(), O.load_r8b_ref()
    if ref is None or r8 is None:
        raise SystemExit("oracle/_ref is not built: run `make -C oracle` where /root/reference exists")
    # ---- r8brain
    cases = {}
    for tag, ch, n, src, dst, seed in (("c5_stereo_0p1s", 2, 9600, 96000.0, 48000.0, 21), ("c5_mono_ragged", 1, 20001, 96000.0, 48000.0, 22),
                                       ("c5_stereo_0p25s", 2, 24000, 96000.0, 48000.0, 23), ("c5_three_short", 3, 1000, 96000.0, 48000.0, 24),
                                       ("up_32_48", 2, 3200, 32000.0, 48000.0, 25), ("down_144_48", 2, 14400, 144000.0, 48000.0, 26),
                                       ("down_72_48", 1, 7201, 72000.0, 48000.0, 27), ("up_48_96", 2, 4800, 48000.0, 96000.0, 28),
                                       ("up_16_48", 1, 1601, 16000.0, 48000.0, 29), ("down_64_48", 2, 6400, 64000.0, 48000.0, 30),
                                       ("ms_441_48", 2, 4410, 44100.0, 48000.0, 31), ("ms_48_441", 2, 4800, 48000.0, 44100.0, 32),
                                       ("hb_48_192", 2, 1200, 48000.0, 192000.0, 33), ("hb_192_48", 2, 4800, 192000.0, 48000.0, 34),
                                       ("hb_192_441", 1, 4800, 192000.0, 44100.0, 35), ("hb_8_96", 1, 400, 8000.0, 96000.0, 36),
                                       ("hb_96_16", 2, 2400, 96000.0, 16000.0, 37),
                                       ("sp_441_48001", 3, 2500, 44100.0, 48001.0, 38), ("sp_48_50854", 2, 3000, 48000.0, float(np.float32(50854.3)), 39),
                                       ("sp_441_14000", 2, 4000, 44100.0, float(np.float32(14000.3)), 40)):
        x = O.noise(ch, n, seed)
        cases[tag + "_x"] = x
        cases[tag + "_y"] = r8b(r8, x, src, dst)
        cases[tag + "_rates"] = np.array([src, dst], np.float64)
    np.savez_compressed(os.path.join(HERE, "r8brain.npz"), **cases)
    # ---- phase_vocoder / inverse_phase_vocoder
    rng = np.random.default_rng(20260104)
    pv_cases = {}
    for hop, dft in ((512, 2048), (128, 4096), (256, 1024)):
        n = 8000
        sr = np.float32(48000.0)
        ar = np.float32(sr / np.float32(hop))
        re = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 3, n)).astype(np.float32)
        im = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 3, n)).astype(np.float32)
        re[:300] = 0; im[:150] = 0; im[300:450] = 0; re[600:650] = 1e-42; im[650:700] = -1e-42
        bins = rng.integers(0, dft // 2 + 1, n)
        binf = (bins.astype(np.float32) * sr / np.float32(dft)).astype(np.float32)
        prev = np.float32(rng.uniform(-np.pi, np.pi, n)).astype(np.float64)
        prev[:1000] = 0.0
        state = prev.copy()
        m = np.empty(n, np.float32); f = np.empty(n, np.float32)
        ref.ref_phase_vocoder_batch(n, state, re, im, binf, ar, sr, m, f)
        key = "h%d_d%d_" % (hop, dft)
        pv_cases.update({key + "re": re, key + "im": im, key + "binf": binf, key + "prev": prev, key + "m": m, key + "f": f, key + "state": state,
                         key + "rates": np.array([ar, sr], np.float32)})
        # inverse: running phases incl. negative and large ones
        ph = rng.uniform(-50.0, 2000.0, n)
        ph[:1000] = rng.uniform(0, 6.0, 1000)
        mm = (10 ** rng.uniform(-5, 3, n)).astype(np.float32)
        ff = rng.uniform(-100.0, 24000.0, n).astype(np.float32)
        st = ph.copy()
        xr = np.empty(n, np.float32); xi = np.empty(n, np.float32)
        ref.ref_inverse_phase_vocoder_batch(n, st, mm, ff, ar, xr, xi)
        pv_cases.update({key + "inv_ph": ph, key + "inv_m": mm, key + "inv_f": ff, key + "inv_state": st, key + "inv_re": xr, key + "inv_im": xi})
    np.savez_compressed(os.path.join(HERE, "phase_vocoder.npz"), **pv_cases)
    # ---- hann
    hann = {"W%d" % W: np.array([ref.ref_hann(np.float32(i) / np.float32(W - 1)) for i in range(W)], np.float32) for W in (64, 256, 1000, 1024, 2048, 4096)}
    xs = rng.random(4000).astype(np.float32)
    hann["xs"] = xs
    hann["at_xs"] = np.array([ref.ref_hann(float(x)) for x in xs], np.float32)
    np.savez_compressed(os.path.join(HERE, "hann.npz"), **hann)
    # ---- PVBuffer: unit conversions + a .flan file written by the reference's own save()
    fmt = O.RefPVFormat(2, 9, 65, 48000.0, 48000.0 / 32, 128)
    vals = rng.uniform(0, 100, 200).astype(np.float32)
    conv = {"format": np.array([2, 9, 65, 48000.0, 48000.0 / 32, 128], np.float64), "vals": vals,
            "hop_size": np.int64(ref.ref_pv_hop_size(fmt)), "dft_size": np.int64(ref.ref_pv_dft_size(fmt))}
    for name in ("ref_pv_bin_to_frequency", "ref_pv_frequency_to_bin", "ref_pv_time_to_frame", "ref_pv_frame_to_time"):
        conv[name[7:]] = np.array([getattr(ref, name)(fmt, float(v)) for v in vals], np.float32)
    mf = np.stack([rng.uniform(0, 128, (2, 9, 65)), rng.uniform(-200, 24000, (2, 9, 65))], -1).astype(np.float32)
    mf[0, 0, 0] = (300.0, 50000.0)                                # clamps to +-1 before quantisation (PVBuffer.cpp:113-114)
    mf[1, 2, 3] = (-5.0, -60000.0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ref.flan").encode()
        assert ref.ref_pv_save(fmt, mf.reshape(-1), path) == 1
        conv["flan_file"] = np.frombuffer(open(path, "rb").read(), np.uint8)
        got = O.RefPVFormat()
        back = np.zeros(mf.size, np.float32)
        assert ref.ref_pv_load(path, C.byref(got), back.ctypes.data_as(C.c_void_p), back.size) == 1
        conv["flan_loaded"] = back.reshape(mf.shape)
        conv["flan_loaded_format"] = np.array([got.num_channels, got.num_frames, got.num_bins, got.sample_rate, got.analysis_rate, got.window_size], np.float64)
    conv["mf"] = mf
    np.savez_compressed(os.path.join(HERE, "pvbuffer.npz"), **conv)
    # ---- interpolators
    grid = np.linspace(0, 1, 2001, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "interpolators.npz"), grid=grid,
                        **{"kind%d" % k: np.array([ref.ref_interpolate(k, float(x)) for x in grid], np.float32) for k in range(9)})
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
