"""ctypes doorway to the CPU checker under oracle/ (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The product
(flan_amd/, include/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = os.path.join(ORACLE_DIR, "liboracle.so")
_REF = os.path.join(ORACLE_DIR, "_ref", "libflanref.so")
_R8B = os.path.join(ORACLE_DIR, "_ref", "libr8bref.so")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build():
    """(Re)build the checker; cheap when up to date."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    lib.oracle_pi2.restype = C.c_float
    lib.oracle_hann.restype = C.c_float
    lib.oracle_hann.argtypes = [C.c_float]
    lib.oracle_hann_window.argtypes = [f32p, C.c_int]
    lib.oracle_phase_vocoder.argtypes = [C.POINTER(C.c_double), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.oracle_inverse_phase_vocoder.argtypes = [C.POINTER(C.c_double), C.c_float, C.c_float, C.c_float,
                                                 C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for name in ("oracle_bin_to_frequency", "oracle_frequency_to_bin", "oracle_time_to_frame", "oracle_frame_to_time"):
        fn = getattr(lib, name)
        fn.restype = C.c_float
        fn.argtypes = [C.c_float, C.c_float, C.c_int]
    lib.oracle_r2c.argtypes = [f32p, C.c_int, f32p]
    lib.oracle_c2r.argtypes = [f32p, C.c_int, f32p]
    lib.oracle_num_pv_frames.restype = C.c_int64
    lib.oracle_num_pv_frames.argtypes = [C.c_int64, C.c_int]
    lib.oracle_analyze.argtypes = [f32p, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_int, f32p]
    lib.oracle_hop_size.argtypes = [C.c_float, C.c_float]
    lib.oracle_synthesize.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int, f32p]
    lib.oracle_modify_time_out_frames.restype = C.c_int64
    lib.oracle_modify_time_out_frames.argtypes = [f32p, C.c_int64, C.c_int, C.c_float, C.c_int]
    lib.oracle_modify_time.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, f32p, C.c_int64, f32p]
    lib.oracle_modify_time_interp.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, f32p, C.c_int64, C.c_int, f32p]
    lib.oracle_modify_frequency_interp.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, f32p, f32p, C.c_int, f32p]
    lib.oracle_stretch_map.argtypes = [f32p, C.c_int64, C.c_int, C.c_float, C.c_int]
    lib.oracle_modify_frequency.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, f32p, f32p, f32p]
    lib.oracle_repitch_map.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, f32p, f32p]
    lib.oracle_shape_affine.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_int, f32p]
    lib.oracle_mid_side.argtypes = [f32p, C.c_int64, f32p]
    lib.oracle_noise.argtypes = [f32p, C.c_int, C.c_int64, C.c_uint32]
    lib.oracle_r8b_default_lowpass_half.argtypes = [f64p, C.c_int]
    lib.oracle_resample_out_frames.restype = C.c_int64
    lib.oracle_resample_out_frames.argtypes = [C.c_int64, C.c_float, C.c_float]
    lib.oracle_resample_2to1.argtypes = [f32p, C.c_int64, f32p, C.c_int64]
    lib.oracle_resample_rational.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_int, C.c_int]
    lib.oracle_r8b_default_lowpass.argtypes = [C.c_double, C.c_double, f64p, C.c_int]
    ip = C.POINTER(C.c_int)
    lib.oracle_resample_two_stage_shape.argtypes = [C.c_double, C.c_double, ip, C.POINTER(C.c_double), ip, ip, ip]
    lib.oracle_r8b_frac_bank.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.oracle_resample_two_stage.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_double, C.c_double]
    dp = C.POINTER(C.c_double)
    lib.oracle_resample_chain_shape.argtypes = [C.c_double, C.c_double, ip, ip, ip, dp, dp, ip, ip, ip, ip, ip, ip]
    lib.oracle_resample_chain.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_double, C.c_double, C.c_int64]
    lib.oracle_resample_general.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.c_double, C.c_double, C.c_int64]
    lib.oracle_resample_stage_list.argtypes = [C.c_double, C.c_double, C.c_char_p, C.c_int]
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    lib.oracle_interpolate.restype = C.c_float
    lib.oracle_interpolate.argtypes = [C.c_int, C.c_float]
    for name in ("oracle_replace_amplitudes", "oracle_subtract_amplitudes"):
        getattr(lib, name).argtypes = [f32p, C.c_int, C.c_int64, C.c_int, f32p, C.c_int, C.c_int64, C.c_int, f32p, f32p]
    lib.oracle_resonate_out_frames.restype = C.c_int64
    lib.oracle_resonate_out_frames.argtypes = [C.c_int64, C.c_float, C.c_float, C.c_int]
    lib.oracle_resonate.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_int64, f32p, C.c_int, f32p]
    lib.oracle_n_loudest_partials.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, i32p, C.c_int, f32p]
    lib.oracle_desample.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, f32p, C.c_int, f32p]
    lib.oracle_time_extrapolate.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int64, C.c_int64, C.c_int64, f32p, f32p]
    lib.oracle_get_frame.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, f32p]
    lib.oracle_select.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, f32p, C.c_int64, f32p]
    lib.oracle_freeze_plan.restype = C.c_int64
    lib.oracle_freeze_plan.argtypes = [C.c_int64, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.oracle_select_frames.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, i32p, C.c_int64, f32p]
    lib.oracle_cut_frames_range.restype = None
    lib.oracle_cut_frames_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.oracle_place_frames.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, f32p, C.c_int, C.c_int64, C.c_int, C.c_int64]
    lib.oracle_harmonic_scale.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, f32p, C.c_int, C.c_int, f32p]
    lib.oracle_modify_out_frames.restype = C.c_int64
    lib.oracle_modify_out_frames.argtypes = [f32p, C.c_int64, C.c_int, C.c_float, C.c_int]
    lib.oracle_modify.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, f32p, f32p, C.c_int, C.c_int64, f32p]
    lib.oracle_spline.restype = None
    lib.oracle_spline.argtypes = [f64p, f64p, C.c_int, f64p, C.c_int, f64p]
    lib.oracle_stretch_spline_out_frames.restype = C.c_int64
    lib.oracle_stretch_spline_out_frames.argtypes = [C.c_void_p, C.c_int64]
    lib.oracle_stretch_spline.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64, f32p]
    lib.oracle_smear_time_plan.restype = None
    lib.oracle_smear_time_plan.argtypes = [C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.oracle_smear_time.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_int64, f32p]
    return lib


lib = _load()


class RefPVFormat(C.Structure):
    _fields_ = [("num_channels", C.c_int32), ("num_frames", C.c_int32), ("num_bins", C.c_int32),
                ("sample_rate", C.c_float), ("analysis_rate", C.c_float), ("window_size", C.c_int32)]


def load_ref():
    """The real reference TUs (oracle/_ref/libflanref.so); None when it was never built (no /root/reference)."""
    if not os.path.exists(_REF):
        return None
    ref = C.CDLL(_REF)
    ref.ref_hann.restype = C.c_float
    ref.ref_hann.argtypes = [C.c_float]
    ref.ref_pi2.restype = C.c_float
    ref.ref_phase_vocoder_batch.argtypes = [C.c_int64, f64p, f32p, f32p, f32p, C.c_float, C.c_float, f32p, f32p]
    ref.ref_inverse_phase_vocoder_batch.argtypes = [C.c_int64, f64p, f32p, f32p, C.c_float, f32p, f32p]
    ref.ref_pv_hop_size.argtypes = [RefPVFormat]
    ref.ref_pv_dft_size.argtypes = [RefPVFormat]
    for name in ("ref_pv_bin_to_frequency", "ref_pv_frequency_to_bin", "ref_pv_time_to_frame", "ref_pv_frame_to_time"):
        fn = getattr(ref, name)
        fn.restype = C.c_float
        fn.argtypes = [RefPVFormat, C.c_float]
    ref.ref_pv_buffer_pos.restype = C.c_int64
    ref.ref_pv_buffer_pos.argtypes = [RefPVFormat, C.c_int, C.c_int, C.c_int]
    ref.ref_pv_is_nan_or_inf.argtypes = [RefPVFormat, f32p]
    ref.ref_pv_save.argtypes = [RefPVFormat, f32p, C.c_char_p]
    ref.ref_pv_load.argtypes = [C.c_char_p, C.POINTER(RefPVFormat), C.c_void_p, C.c_int64]
    ref.ref_interpolate.restype = C.c_float
    ref.ref_interpolate.argtypes = [C.c_int, C.c_float]
    if hasattr(ref, "ref_function_sample2d"):
        ref.ref_function_sample2d.restype = C.c_int64
        ref.ref_function_sample2d.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int] + [C.c_float] * 6 + [C.c_int, f32p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        ref.ref_function_sample1d.restype = C.c_int64
        ref.ref_function_sample1d.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, f32p, C.c_int64, C.POINTER(C.c_int)]
    if hasattr(ref, "ref_spline"):
        ref.ref_spline.restype = None
        ref.ref_spline.argtypes = [f64p, f64p, C.c_int, f64p, C.c_int, f64p]
    return ref


# ----------------------------------------------------------------------------------------------------------
# numpy-level helpers
# ----------------------------------------------------------------------------------------------------------

def hann_window(window):
    w = np.empty(window, np.float32)
    lib.oracle_hann_window(w, window)
    return w


def num_pv_frames(n, hop):
    return int(lib.oracle_num_pv_frames(n, hop))


def analyze(audio, sample_rate, window=2048, hop=128, dft=4096):
    """audio: float32 [ch][n] -> MF float32 [ch][F][bins][2]"""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    F = num_pv_frames(n, hop)
    out = np.empty((ch, F, dft // 2 + 1, 2), np.float32)
    rc = lib.oracle_analyze(audio, ch, n, sample_rate, window, hop, dft, out.reshape(-1))
    assert rc == 0, rc
    return out


def synthesize(pv, sample_rate, analysis_rate, window):
    """pv: float32 [ch][F][bins][2] -> (audio float32 [ch][F*hop], nan_flag)"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    hop = lib.oracle_hop_size(sample_rate, analysis_rate)
    out = np.empty((ch, F * hop), np.float32)
    rc = lib.oracle_synthesize(pv.reshape(-1), ch, F, bins, sample_rate, analysis_rate, window, out.reshape(-1))
    assert rc >= 0, rc
    return out, rc


def stretch_map(factor_grid, sample_rate, hop):
    g = np.ascontiguousarray(factor_grid, np.float32).copy()
    F, bins = g.shape
    lib.oracle_stretch_map(g.reshape(-1), F, bins, sample_rate, hop)
    return g


def modify_time(pv, sample_rate, hop, mod_seconds, interp=0):
    pv = np.ascontiguousarray(pv, np.float32)
    mod = np.ascontiguousarray(mod_seconds, np.float32)
    ch, F, bins, _ = pv.shape
    Fo = int(lib.oracle_modify_time_out_frames(mod.reshape(-1), F, bins, sample_rate, hop))
    out = np.empty((ch, max(Fo, 0), bins, 2), np.float32)
    if Fo > 0:
        lib.oracle_modify_time_interp(pv.reshape(-1), ch, F, bins, sample_rate, hop, mod.reshape(-1), Fo, interp, out.reshape(-1))
    return out


def stretch(pv, sample_rate, hop, factor_grid, interp=0):
    return modify_time(pv, sample_rate, hop, stretch_map(factor_grid, sample_rate, hop), interp)


def repitch_map(pv, sample_rate, factor_grid):
    pv = np.ascontiguousarray(pv, np.float32)
    g = np.ascontiguousarray(factor_grid, np.float32).copy()
    ch, F, bins, _ = pv.shape
    inmod = np.empty((ch, F, bins), np.float32)
    lib.oracle_repitch_map(pv.reshape(-1), ch, F, bins, sample_rate, g.reshape(-1), inmod.reshape(-1))
    return g, inmod


def modify_frequency(pv, sample_rate, mod_hz, in_modified, interp=0):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    out = np.empty_like(pv)
    lib.oracle_modify_frequency_interp(pv.reshape(-1), ch, F, bins, sample_rate,
                                       np.ascontiguousarray(mod_hz, np.float32).reshape(-1),
                                       np.ascontiguousarray(in_modified, np.float32).reshape(-1), interp, out.reshape(-1))
    return out


def repitch(pv, sample_rate, factor_grid, interp=0):
    g, inmod = repitch_map(pv, sample_rate, factor_grid)
    return modify_frequency(pv, sample_rate, g, inmod, interp)


def shape_affine(pv, sample_rate, a, b, c, d, use_shift_alignment=False):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    out = np.empty_like(pv)
    lib.oracle_shape_affine(pv.reshape(-1), ch, F, bins, sample_rate, a, b, c, d, int(use_shift_alignment), out.reshape(-1))
    return out


def sample_grid(fn, F, bins, analysis_rate, bin_width):
    """The grid the checker's frame processors take for a user function of ( time, frequency ): [frame][bin], the argument of point ( x, y ) being
    ( x * float32( 1 / analysis_rate ), y * bin_width ) with every product rounded to fp32 -- PV::sample_function_over_domain (PV/PV.h:31-35) through
    Function::sample (Function.h:155-171).  tests/test_ref_made_golden.py holds this convention to grids the reference's own header made."""
    xs = np.float32(1.0) / np.float32(analysis_rate)
    t = (np.arange(F, dtype=np.float32) * xs).astype(np.float32)
    f = (np.arange(bins, dtype=np.float32) * np.float32(bin_width)).astype(np.float32)
    return np.asarray(fn(t[:, None], f[None, :]), np.float32).reshape(F, bins)


def _grid(g, F, bins):
    """a sampled user function as a full float32 [F][bins] grid (scalars are broadcast, like a constant Function)"""
    if np.isscalar(g):
        return np.full((F, bins), g, np.float32)
    g = np.ascontiguousarray(g, np.float32)
    assert g.shape == (F, bins), (g.shape, F, bins)
    return g


def _combine(fn, pv, src, amount):
    pv = np.ascontiguousarray(pv, np.float32)
    src = np.ascontiguousarray(src, np.float32)
    ch, F, bins, _ = pv.shape
    sch, sF, sbins, _ = src.shape
    out = np.empty_like(pv)
    fn(pv.reshape(-1), ch, F, bins, src.reshape(-1), sch, sF, sbins, _grid(amount, F, bins).reshape(-1), out.reshape(-1))
    return out


def replace_amplitudes(pv, src, amount):
    return _combine(lib.oracle_replace_amplitudes, pv, src, amount)


def subtract_amplitudes(pv, src, amount):
    return _combine(lib.oracle_subtract_amplitudes, pv, src, amount)


def resonate(pv, sample_rate, hop, length_seconds, decay, pow_mode=0):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    Fo = int(lib.oracle_resonate_out_frames(F, length_seconds, sample_rate, hop))
    out = np.empty((ch, Fo, bins, 2), np.float32)
    lib.oracle_resonate(pv.reshape(-1), ch, F, bins, sample_rate, hop, Fo, _grid(decay, Fo, bins).reshape(-1), pow_mode, out.reshape(-1))
    return out


def n_loudest_partials(pv, n, remove=False):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    n = np.full(F, n, np.int32) if np.isscalar(n) else np.ascontiguousarray(n, np.int32)
    out = np.empty_like(pv)
    lib.oracle_n_loudest_partials(pv.reshape(-1), ch, F, bins, n, int(remove), out.reshape(-1))
    return out


def desample(pv, ratio, interp=0):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    out = np.empty_like(pv)
    lib.oracle_desample(pv.reshape(-1), ch, F, bins, _grid(ratio, F, bins).reshape(-1), interp, out.reshape(-1))
    return out


def time_extrapolate(pv, sample_rate, start_frame, end_frame, out_frames, interp_samples):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    samples = np.ascontiguousarray(interp_samples, np.float32)
    assert samples.shape == (out_frames - start_frame,)
    out = np.empty((ch, out_frames, bins, 2), np.float32)
    lib.oracle_time_extrapolate(pv.reshape(-1), ch, F, bins, sample_rate, start_frame, end_frame, out_frames, samples, out.reshape(-1))
    return out


def time_extrapolate_interp_samples(start_frame, end_frame, out_frames, interp=0):
    """PVModify.cpp:631-633 literally (the sample for output frame start+k is interp((k - start)/(end - start)))"""
    k = np.arange(out_frames - start_frame, dtype=np.int64)
    x = (k - start_frame).astype(np.float32) / np.float32(end_frame - start_frame)
    return np.array([lib.oracle_interpolate(interp, float(v)) for v in x], np.float32)


def get_frame(pv, frame_pos, interp=0):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    out = np.empty((ch, 1, bins, 2), np.float32)
    assert lib.oracle_get_frame(pv.reshape(-1), ch, F, bins, frame_pos, interp, out.reshape(-1)) == 0
    return out


def select(pv, sample_rate, hop, selector_tf):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    sel = np.ascontiguousarray(selector_tf, np.float32)
    Fo = sel.shape[0]
    out = np.empty((ch, Fo, bins, 2), np.float32)
    lib.oracle_select(pv.reshape(-1), ch, F, bins, sample_rate, hop, sel.reshape(-1), Fo, out.reshape(-1))
    return out


def freeze_plan(num_frames, sample_rate, hop, times, lengths):
    times = np.ascontiguousarray(times, np.float32)
    lengths = np.ascontiguousarray(lengths, np.float32)
    n = len(times)
    tp = times.ctypes.data if n else None
    lp = lengths.ctypes.data if n else None
    Fo = lib.oracle_freeze_plan(num_frames, sample_rate, hop, tp, lp, n, None)
    src = np.empty(Fo, np.int32)
    lib.oracle_freeze_plan(num_frames, sample_rate, hop, tp, lp, n, src.ctypes.data)
    return src


def select_frames(pv, src_frames):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    src = np.ascontiguousarray(src_frames, np.int32)
    out = np.empty((ch, len(src), bins, 2), np.float32)
    lib.oracle_select_frames(pv.reshape(-1), ch, F, bins, src, len(src), out.reshape(-1))
    return out


def freeze(pv, sample_rate, hop, times, lengths):
    return select_frames(pv, freeze_plan(np.shape(pv)[1], sample_rate, hop, times, lengths))


def cut_frames(pv, start, end):
    pv = np.ascontiguousarray(pv, np.float32)
    s, c = C.c_int32(0), C.c_int32(0)
    lib.oracle_cut_frames_range(pv.shape[1], start, end, C.byref(s), C.byref(c))
    if c.value <= 0:
        return None
    return select_frames(pv, np.arange(s.value, s.value + c.value, dtype=np.int32))


def join(pvs):
    pvs = [np.ascontiguousarray(p, np.float32) for p in pvs]
    ch, _, bins, _ = pvs[0].shape
    Fo = sum(p.shape[1] for p in pvs)
    out = np.zeros((ch, Fo, bins, 2), np.float32)
    at = 0
    for p in pvs:
        assert lib.oracle_place_frames(p.reshape(-1), p.shape[0], p.shape[1], p.shape[2], out.reshape(-1), ch, Fo, bins, at) == 0
        at += p.shape[1]
    return out


def harmonic_scale(pv, sample_rate, series, mode):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    series = np.ascontiguousarray(series, np.float32)
    H = series.shape[1]
    out = np.empty_like(pv)
    lib.oracle_harmonic_scale(pv.reshape(-1), ch, F, bins, sample_rate, series.reshape(-1) if H else np.zeros(1, np.float32), H, mode, out.reshape(-1))
    return out


def modify_out_frames(mod_tf, sample_rate, hop):
    mod = np.ascontiguousarray(mod_tf, np.float32)
    return int(lib.oracle_modify_out_frames(mod.reshape(-1), mod.shape[0], mod.shape[1], sample_rate, hop))


def modify(pv, sample_rate, hop, mod_tf, in_f, interp=0, out_frames=None):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    mod = np.ascontiguousarray(mod_tf, np.float32)
    in_f = np.ascontiguousarray(in_f, np.float32)
    Fo = modify_out_frames(mod, sample_rate, hop) if out_frames is None else out_frames
    if Fo <= 0:
        return None
    out = np.empty((ch, Fo, bins, 2), np.float32)
    lib.oracle_modify(pv.reshape(-1), ch, F, bins, sample_rate, hop, mod.reshape(-1), in_f.reshape(-1), interp, Fo, out.reshape(-1))
    return out


def spline(x, y, t):
    x, y, t = (np.ascontiguousarray(v, np.float64) for v in (x, y, t))
    out = np.empty(len(t), np.float64)
    lib.oracle_spline(x, y, len(x), t, len(t), out)
    return out


def stretch_spline(pv, steps):
    """PV::stretch_spline; steps: uint32 [F-1] = the safeInterpolation value of every frame but the last"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    steps = np.ascontiguousarray(steps, np.uint32)
    assert steps.shape == (F - 1,)
    Fo = int(lib.oracle_stretch_spline_out_frames(steps.ctypes.data, F))
    out = np.empty((ch, Fo, bins, 2), np.float32)
    assert lib.oracle_stretch_spline(pv.reshape(-1), ch, F, bins, steps.ctypes.data, Fo, out.reshape(-1)) == 0
    return out


def smear_time_plan(num_frames, num_bins, sample_rate, hop, smear):
    left, Fo, half = C.c_int32(0), C.c_int64(0), C.c_int32(0)
    if np.isscalar(smear):
        lib.oracle_smear_time_plan(num_frames, num_bins, sample_rate, hop, None, float(smear), C.byref(left), C.byref(Fo), C.byref(half))
    else:
        g = np.ascontiguousarray(smear, np.float32)
        lib.oracle_smear_time_plan(num_frames, num_bins, sample_rate, hop, g.ctypes.data, 0.0, C.byref(left), C.byref(Fo), C.byref(half))
    return left.value, Fo.value, half.value


def smear_time(pv, sample_rate, hop, smear, granularity, dist, true_left, out_frames):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    sg = None if np.isscalar(smear) else np.ascontiguousarray(smear, np.float32)
    gg = None if np.isscalar(granularity) else np.ascontiguousarray(granularity, np.int32)
    dist = np.ascontiguousarray(dist, np.float32)
    out = np.empty((ch, out_frames, bins, 2), np.float32)
    lib.oracle_smear_time(pv.reshape(-1), ch, F, bins, sample_rate, hop, sg.ctypes.data if sg is not None else None, 0.0 if sg is not None else float(smear),
                          gg.ctypes.data if gg is not None else None, 0 if gg is not None else int(granularity),
                          dist.ctypes.data if len(dist) else None, len(dist), true_left, out_frames, out.reshape(-1))
    return out


def smear_distribution(dist_samples_2, fn=None):
    """PVModify.cpp:558-560: distribution( x / dist_samples_2 ) for x in [-dist_samples_2, dist_samples_2); the default
    distribution is PV.h:336-339, 0.5 ( 1 + cos( pi t ) ) in double, rounded to float"""
    if dist_samples_2 <= 0:
        return np.zeros(0, np.float32)
    x = np.arange(-dist_samples_2, dist_samples_2, dtype=np.int64)
    t = (x.astype(np.float32) * np.float32(np.float32(1.0) / np.float32(dist_samples_2))).astype(np.float32)
    if fn is None:
        return (0.5 * (1.0 + np.cos(np.pi * t.astype(np.float64)))).astype(np.float32)
    return np.array([fn(float(v)) for v in t], np.float32)


def mid_side(audio):
    audio = np.ascontiguousarray(audio, np.float32)
    assert audio.shape[0] == 2
    out = np.empty_like(audio)
    lib.oracle_mid_side(audio.reshape(-1), audio.shape[1], out.reshape(-1))
    return out


def resample_2to1(audio, src_rate, dst_rate):
    """Audio::resample for src = 2 dst: [ch][n] -> [ch][n_out], the whole buffer as one stream"""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    n_out = int(lib.oracle_resample_out_frames(n, src_rate, dst_rate))
    out = np.empty((ch, n_out), np.float32)
    rc = lib.oracle_resample_2to1(audio.reshape(-1), ch * n, out.reshape(-1), ch * n_out)
    assert rc == 0
    return out


# the ratios r8brain serves with ONE block convolver (CDSPResampler.h:139-207): (src, dst) -> (up, down)
SINGLE_STEP_RATIOS = {(2, 1): (1, 2), (3, 1): (1, 3), (3, 2): (2, 3), (2, 3): (3, 2), (4, 3): (3, 4), (1, 2): (2, 1), (1, 3): (3, 1)}


def resample_rational(audio, src_rate, dst_rate, up, down):
    """Audio::resample for a single-step ratio: [ch][n] -> [ch][n_out], the whole buffer as one stream"""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    n_out = int(lib.oracle_resample_out_frames(n, src_rate, dst_rate))
    out = np.empty((ch, n_out), np.float32)
    rc = lib.oracle_resample_rational(audio.reshape(-1), ch * n, out.reshape(-1), ch * n_out, up, down)
    assert rc == 0
    return out


def two_stage_shape(src_rate, dst_rate):
    """None, or what CDSPResampler( src, dst ) builds when it is one block convolver + one whole-stepping interpolator"""
    up, third, ins, outs, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
    if not lib.oracle_resample_two_stage_shape(src_rate, dst_rate, C.byref(up), C.byref(nf), C.byref(third), C.byref(ins), C.byref(outs)):
        return None
    return dict(up=up.value, norm_freq=nf.value, third=bool(third.value), in_step=ins.value, out_step=outs.value)


def frac_bank(fracs, third=False):
    """the whole-stepping bank of fractional-delay filters: [fracs][flt_len] fp64"""
    flt_len = lib.oracle_r8b_frac_bank(fracs, int(third), None, 0)
    rows = np.zeros((fracs, flt_len))
    lib.oracle_r8b_frac_bank(fracs, int(third), rows.ctypes.data, rows.size)
    return rows


def resample_two_stage(audio, src_rate, dst_rate):
    """Audio::resample for a block convolver + whole-stepping interpolator ratio (44.1 <-> 48 kHz ...): [ch][n] -> [ch][n_out]"""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    n_out = int(lib.oracle_resample_out_frames(n, src_rate, dst_rate))
    out = np.empty((ch, n_out), np.float32)
    rc = lib.oracle_resample_two_stage(audio.reshape(-1), ch * n, out.reshape(-1), ch * n_out, float(src_rate), float(dst_rate))
    assert rc == 0, "not a two-stage ratio"
    return out


def chain_shape(src_rate, dst_rate):
    """None, or the chain CDSPResampler( src, dst ) builds when the checker restates it: [half-band downsamplers] -> block convolver
    -> [half-band upsamplers] -> [whole-stepping interpolator]"""
    iv = [C.c_int() for _ in range(9)]
    nf, gain = C.c_double(), C.c_double()
    if not lib.oracle_resample_chain_shape(src_rate, dst_rate, C.byref(iv[0]), C.byref(iv[1]), C.byref(iv[2]), C.byref(nf), C.byref(gain),
                                           C.byref(iv[3]), C.byref(iv[4]), C.byref(iv[5]), C.byref(iv[6]), C.byref(iv[7]), C.byref(iv[8])):
        return None
    d = dict(hb_down=iv[0].value, up=iv[1].value, down=iv[2].value, norm_freq=nf.value, gain=gain.value, hb_up=iv[3].value,
             third=bool(iv[4].value), interp=bool(iv[5].value), in_step=iv[6].value, out_step=iv[7].value)
    if iv[8].value:
        d["spline"] = True
    return d


def resample_chain(audio, src_rate, dst_rate):
    """Audio::resample through any chain the checker restates: [ch][n] -> [ch][n_out], the whole buffer as one stream"""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    n_out = int(lib.oracle_resample_out_frames(n, src_rate, dst_rate))
    out = np.empty((ch, n_out), np.float32)
    rc = lib.oracle_resample_chain(audio.reshape(-1), ch * n, out.reshape(-1), ch * n_out, float(src_rate), float(dst_rate), n)   # oneshot feeds n samples per call
    assert rc == 0, "chain not restated"
    return out


def resample_stages(src_rate, dst_rate):
    """the chain CDSPResampler( src, dst ) builds, as text ("conv:2/1@0.5,tb2,g2 frac:147/80 "), or None when the checker does not restate it"""
    buf = C.create_string_buffer(1024)
    return buf.value.decode() if lib.oracle_resample_stage_list(float(src_rate), float(dst_rate), buf, 1024) else None


def resample_general(audio, src_rate, dst_rate):
    """Audio::resample through whatever chain r8brain builds (the stage-list form of the checker): [ch][n] -> [ch][n_out]"""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    n_out = int(lib.oracle_resample_out_frames(n, src_rate, dst_rate))
    out = np.empty((ch, n_out), np.float32)
    rc = lib.oracle_resample_general(audio.reshape(-1), ch * n, out.reshape(-1), ch * n_out, float(src_rate), float(dst_rate), n)
    assert rc == 0, "chain not restated"
    return out


def load_r8b_ref():
    """the real r8brain resampler the reference vendors (oracle/_ref/libr8bref.so); None when it was never built"""
    if not os.path.exists(_R8B):
        return None
    ref = C.CDLL(_R8B)
    ref.ref_r8b_resample.argtypes = [f32p, C.c_int, C.c_double, C.c_double, C.c_int, f32p, C.c_int]
    return ref


def noise(ch, n, seed=1234):
    out = np.empty((ch, n), np.float32)
    lib.oracle_noise(out.reshape(-1), ch, n, seed)
    return out


def sine(n, freq=440.0, amp=0.5, sr=48000.0):
    """SURVEY 8c anchor signal: computed in double, rounded once."""
    t = np.arange(n, dtype=np.float64)
    return (amp * np.sin(2.0 * np.pi * freq * t / sr)).astype(np.float32)[None, :]
