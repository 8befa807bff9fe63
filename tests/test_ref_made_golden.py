"""The oracle against vectors the REFERENCE ITSELF produced (tests/golden/ref_made/*.npz, made by make_ref_made.py from the
reference translation units that build unmodified: phase_vocoder.cpp, WindowFunctions.cpp, PV/PVBuffer.cpp, Utility/Interpolator.cpp
and the vendored r8brain).  Unlike tests/test_oracle_vs_ref.py these need neither /root/reference nor the prebuilt oracle/_ref:
the pinned part of the oracle travels as data.  Bit-exact unless said otherwise."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_made")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_hann_against_reference_made():
    g = np.load(os.path.join(G, "hann.npz"))
    for W in (64, 256, 1000, 1024, 2048, 4096):
        assert np.array_equal(_bits(O.hann_window(W)), _bits(g["W%d" % W])), W
    got = np.array([O.lib.oracle_hann(float(x)) for x in g["xs"]], np.float32)
    assert np.array_equal(_bits(got), _bits(g["at_xs"]))


@pytest.mark.parametrize("hop,dft", [(512, 2048), (128, 4096), (256, 1024)])
def test_phase_vocoder_against_reference_made(hop, dft):
    z = np.load(os.path.join(G, "phase_vocoder.npz"))
    k = "h%d_d%d_" % (hop, dft)
    g = {name: z[name] for name in z.files if name.startswith(k)}         # (an NpzFile decompresses on every access)
    ar, sr = (float(v) for v in g[k + "rates"])
    re, im, binf, prev = g[k + "re"], g[k + "im"], g[k + "binf"], g[k + "prev"]
    m = C.c_float(); f = C.c_float(); ph = C.c_double()
    bad = 0
    for i in range(len(re)):
        ph.value = prev[i]
        O.lib.oracle_phase_vocoder(C.byref(ph), float(re[i]), float(im[i]), float(binf[i]), ar, sr, C.byref(m), C.byref(f))
        ok = (np.float32(m.value).view(np.uint32) == g[k + "m"][i].view(np.uint32) and
              (np.float32(f.value).view(np.uint32) == g[k + "f"][i].view(np.uint32) or (np.isnan(f.value) and np.isnan(g[k + "f"][i]))) and
              (ph.value == g[k + "state"][i] or (np.isnan(ph.value) and np.isnan(g[k + "state"][i]))))
        bad += not ok
    assert bad == 0
    xr = C.c_float(); xi = C.c_float()
    for i in range(len(re)):
        ph.value = g[k + "inv_ph"][i]
        O.lib.oracle_inverse_phase_vocoder(C.byref(ph), float(g[k + "inv_m"][i]), float(g[k + "inv_f"][i]), ar, C.byref(xr), C.byref(xi))
        bad += not (np.float32(xr.value).view(np.uint32) == g[k + "inv_re"][i].view(np.uint32) and
                    np.float32(xi.value).view(np.uint32) == g[k + "inv_im"][i].view(np.uint32) and ph.value == g[k + "inv_state"][i])
    assert bad == 0


def test_pvbuffer_conversions_against_reference_made():
    g = np.load(os.path.join(G, "pvbuffer.npz"))
    ch, F, bins, sr, ar, W = g["format"]
    hop = O.lib.oracle_hop_size(float(sr), float(ar))
    dft = (int(bins) - 1) * 2
    assert hop == int(g["hop_size"]) and dft == int(g["dft_size"])
    v = g["vals"]
    assert np.array_equal(_bits([O.lib.oracle_bin_to_frequency(float(x), float(sr), dft) for x in v]), _bits(g["bin_to_frequency"]))
    assert np.array_equal(_bits([O.lib.oracle_frequency_to_bin(float(x), float(sr), dft) for x in v]), _bits(g["frequency_to_bin"]))
    assert np.array_equal(_bits([O.lib.oracle_time_to_frame(float(x), float(sr), hop) for x in v]), _bits(g["time_to_frame"]))
    assert np.array_equal(_bits([O.lib.oracle_frame_to_time(float(x), float(sr), hop) for x in v]), _bits(g["frame_to_time"]))


def test_interpolators_against_reference_made():
    g = np.load(os.path.join(G, "interpolators.npz"))
    for kind in range(9):
        got = np.array([O.lib.oracle_interpolate(kind, float(x)) for x in g["grid"]], np.float32)
        assert np.array_equal(_bits(got), _bits(g["kind%d" % kind])), kind


R8B_SINGLE_STEP = ["c5_stereo_0p1s", "c5_mono_ragged", "c5_stereo_0p25s", "c5_three_short", "up_32_48", "down_144_48", "down_72_48", "up_48_96", "up_16_48", "down_64_48"]
UPDOWN = {(96000.0, 48000.0): (1, 2), (32000.0, 48000.0): (3, 2), (144000.0, 48000.0): (1, 3), (72000.0, 48000.0): (2, 3), (48000.0, 96000.0): (2, 1),
          (16000.0, 48000.0): (3, 1), (64000.0, 48000.0): (3, 4)}


@pytest.mark.parametrize("tag", R8B_SINGLE_STEP)
def test_resample_restatement_against_real_r8brain_vectors(tag):
    """oracle/resample_oracle.cpp against what the reference's vendored r8brain produced: >= 99.9 % bit-identical, the rest one ulp"""
    g = np.load(os.path.join(G, "r8brain.npz"))
    x, y = g[tag + "_x"], g[tag + "_y"]
    src, dst = (float(v) for v in g[tag + "_rates"])
    up, down = UPDOWN[(src, dst)]
    got = O.resample_2to1(x, src, dst) if (up, down) == (1, 2) else O.resample_rational(x, src, dst, up, down)
    assert got.shape == y.shape
    same = np.mean(got.view(np.uint32) == y.view(np.uint32))
    worst = np.abs(got.astype(np.float64) - y.astype(np.float64)).max()
    print("\n[%s] bit-identical %.5f  worst %.2e" % (tag, same, worst))
    assert same >= 0.999 and worst <= 1.2e-7            # one fp32 ulp at unit scale (the fp64 sums differ in the 16th digit only)


@pytest.mark.parametrize("tag", ["ms_441_48", "ms_48_441", "hb_48_192", "hb_192_48", "hb_192_441", "hb_8_96", "hb_96_16", "sp_441_48001", "sp_48_50854", "sp_441_14000"])
def test_two_stage_restatement_against_real_r8brain_vectors(tag):
    """44.1 <-> 48 kHz (block convolver + whole-stepping CDSPFracInterpolator) and the half-band chains (4x, 12x up; 4x, 6x down;
    half-band + interpolator) and rates without whole stepping (the spline-interpolated bank, re-based per call): the restatement against
    the vendored r8brain's output"""
    g = np.load(os.path.join(G, "r8brain.npz"))
    x, y = g[tag + "_x"], g[tag + "_y"]
    src, dst = (float(v) for v in g[tag + "_rates"])
    got = O.resample_chain(x, src, dst)
    assert got.shape == y.shape
    same = np.mean(got.view(np.uint32) == y.view(np.uint32))
    worst = np.abs(got.astype(np.float64) - y.astype(np.float64)).max()
    print("\n[%s] bit-identical %.5f  worst %.2e" % (tag, same, worst))
    assert same >= 0.999 and worst <= 1.2e-7


def test_flan_file_written_by_the_reference_loads_here(tmp_path):
    """a .flan image produced by the reference's own PVBuffer::save (Bytes.cpp writeRIFF): our load() gives what the reference's load()
    gave, and our save() of that data reproduces the reference's bytes (host classes: flan_amd/host/PVBuffer.cpp through c_hooks.cpp)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "flan_amd", "host")], check=True)
    C.CDLL(os.path.join(root, "flan_amd", "libflanhip.so"), mode=C.RTLD_GLOBAL)
    host = C.CDLL(os.path.join(root, "flan_amd", "libflan_host.so"))
    host.flan_pv_save_file.argtypes = [O.RefPVFormat, O.f32p, C.c_char_p]
    host.flan_pv_load_file.restype = C.c_int64
    host.flan_pv_load_file.argtypes = [C.c_char_p, C.POINTER(O.RefPVFormat), C.c_void_p, C.c_int64]
    g = np.load(os.path.join(G, "pvbuffer.npz"))
    image = g["flan_file"].tobytes()
    src, dst = str(tmp_path / "ref.flan").encode(), str(tmp_path / "ours.flan").encode()
    open(src, "wb").write(image)
    fmt = O.RefPVFormat()
    mf = np.zeros(g["flan_loaded"].size, np.float32)
    assert host.flan_pv_load_file(src, C.byref(fmt), mf.ctypes.data_as(C.c_void_p), mf.size // 2) == mf.size // 2
    want = g["flan_loaded_format"]
    assert [fmt.num_channels, fmt.num_frames, fmt.num_bins, fmt.sample_rate, fmt.analysis_rate, fmt.window_size] == list(want)
    assert np.array_equal(_bits(mf), _bits(g["flan_loaded"].reshape(-1)))
    # what the reference wrote from the original data, we write from the original data too (bytes 10 / 11 of the RIFF type tag are
    # whatever followed the literal "PV" in the reference's memory: Bytes.cpp writes 4 bytes of a 3-byte literal)
    ch, F, bins, sr, ar, W = g["format"]
    fmt0 = O.RefPVFormat(int(ch), int(F), int(bins), float(sr), float(ar), int(W))
    assert host.flan_pv_save_file(fmt0, np.ascontiguousarray(g["mf"]).reshape(-1), dst) == 1
    ours = open(dst, "rb").read()
    assert len(ours) == len(image) and ours[:10] == image[:10] and ours[12:] == image[12:]


# ---- Function::sample: grids made by the reference's own Function.h / FunctionSample.h (oracle/ref_driver.cpp, compiled unmodified)

def _host_lib():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "flan_amd", "host")], check=True)
    C.CDLL(os.path.join(root, "flan_amd", "libflanhip.so"), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(os.path.join(root, "flan_amd", "libflan_host.so"))
    lib.flan_function_sample2d.restype = C.c_int64
    lib.flan_function_sample2d.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int] + [C.c_float] * 6 + [O.f32p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    lib.flan_function_sample1d.restype = C.c_int64
    lib.flan_function_sample1d.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, O.f32p, C.c_int64, C.POINTER(C.c_int)]
    return lib


def _same_floats(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))))


def test_function_sample_2d_against_reference_made():
    """include/flan/Function.h (the product's host code on the path of configs 3 and 5) against the grids the reference's Function<TF,float>::sample
    produced: element count (the ceil of the extents), small dimension, [x][y] order, the sample points ( x * x_scale, y * y_scale ) for integer x and
    y, the truncated slot arithmetic of fractional bounds (shared and untouched slots included), NaN results, and a constant staying a constant."""
    g = np.load(os.path.join(G, "function_sample.npz"))
    lib = _host_lib()
    for i, (which, a, b, pol, x0, x1, xs, y0, y1, ys, scan) in enumerate(g["cases2d"]):
        n_ref, const_ref, small_ref = (int(v) for v in g["s2d_%d_meta" % i])
        if scan:
            continue                                                                 # (the running sums: next test)
        buf = np.full(1 << 16, -54321.0, np.float32)
        const, small = C.c_int(0), C.c_int64(0)
        n = lib.flan_function_sample2d(int(which), a, b, int(pol), x0, x1, np.float32(xs), y0, y1, np.float32(ys), buf, buf.size, C.byref(const), C.byref(small))
        assert (n, const.value, small.value) == (n_ref, const_ref, small_ref), i
        assert _same_floats(buf[:1 if const_ref else n], g["s2d_%d" % i]), i
    # every execution policy walks the same points from a zero start (the PV methods' call)
    for pol in range(4):
        buf = np.full(1 << 16, -54321.0, np.float32)
        const, small = C.c_int(0), C.c_int64(0)
        n = lib.flan_function_sample2d(0, 3.0, 0.01, pol, 0.0, 37.0, np.float32(1.0 / 93.75), 0.0, 65.0, 375.0, buf, buf.size, C.byref(const), C.byref(small))
        assert n == 2405 and _same_floats(buf[:n], g["s2d_1"]), pol


def test_function_sample_scan_and_the_checkers_grid_convention():
    """(i) PV::stretch's in-place running sum down the frames (PV/PVModify.cpp:376-378) as the reference ran it through FunctionSample2d::at on a
    sampled grid: at( frame, bin ) = vector[ frame * small + bin ] -- the same sums taken on the product's grid in that order land on the same bits;
    (ii) the checker's [frame][bin] grids (oracle_lib.sample_grid: what every frame-processor test hands the oracle and the device) are the
    reference's grid for a PV's domain; (iii) the reference's constant case recorded as it is: at() aliases ONE value, so the running sum doubles it
    once per (frame, bin) step -- 2 -> 2^61 over 15 x 4 steps.  The product deliberately treats a constant like the callable returning it
    (include/flan/Function.h, tests/cpp/host_test.cpp)."""
    g = np.load(os.path.join(G, "function_sample.npz"))
    cases = g["cases2d"]
    plain, scanned = g["s2d_1"].reshape(37, 65), g["s2d_2"].reshape(37, 65)
    assert tuple(cases[1][4:10]) == tuple(cases[2][4:10]) and cases[2][10] == 1
    run = plain.copy()
    for f in range(1, 37):
        run[f] = run[f - 1] + run[f]                                              # fp32, frame by frame, like the reference's +=
    assert _same_floats(run, scanned)
    grid = O.sample_grid(lambda t, f: t * np.float32(3.0) + f * np.float32(0.01), 37, 65, 93.75, 375.0)
    assert _same_floats(grid, plain)
    assert float(g["s2d_10"][0]) == 2.0 * 2.0 ** 60 and tuple(g["s2d_10_meta"]) == (64, 1, 4)
    assert float(g["s2d_11"][0]) == 1.5 * 2.0 ** 4 and tuple(g["s2d_11_meta"]) == (6, 1, 2)


def test_function_sample_1d_against_reference_made():
    g = np.load(os.path.join(G, "function_sample.npz"))
    lib = _host_lib()
    for i, (which, a, b, pol, start, end, scale) in enumerate(g["cases1d"]):
        n_ref, const_ref = (int(v) for v in g["s1d_%d_meta" % i])
        buf = np.full(1 << 12, -54321.0, np.float32)
        const = C.c_int(0)
        n = lib.flan_function_sample1d(int(which), a, b, int(pol), int(start), int(end), np.float32(scale), buf, buf.size, C.byref(const))
        assert (n, const.value) == (n_ref, const_ref), i
        assert _same_floats(buf[:1 if const_ref else n], g["s1d_%d" % i]), i
