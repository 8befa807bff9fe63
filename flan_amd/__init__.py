"""flan_amd -- thin Python doorway to the MI355X phase-vocoder hot path (flan_amd/libflanhip.so).

The product is the HIP library behind the C ABI in include/flanhip.h and the C++ host classes in include/flan/.
This module is plumbing for tests and bench.py: it binds the C ABI with ctypes and nothing else.  There is no
CPU fallback and no import of anything under oracle/: if the library is missing, importing fails; if no GPU is
visible, every compute call raises FlanHipError(FLANHIP_ERR_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLAN_AMD_LIB") or os.path.join(_HERE, "libflanhip.so")   # FLAN_AMD_LIB: A/B a second build

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_CANCELLED, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5


class FlanHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("flanhip error %d: %s" % (code, message))
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError("flan_amd/libflanhip.so is missing: build it with `python flan_amd/build.py` "
                      "(hipcc --offload-arch=gfx950); there is no fallback path")

lib = C.CDLL(LIB_PATH)

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

_SIGS = {
    "flanhip_version": (C.c_int, []),
    "flanhip_last_error": (C.c_char_p, []),
    "flanhip_device_count": (C.c_int, []),
    "flanhip_set_device": (C.c_int, [_i32]),
    "flanhip_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "flanhip_num_pv_frames": (_i64, [_i64, _i32]),
    "flanhip_hop_size": (C.c_int, [_f32, _f32]),
    "flanhip_modify_time_out_frames": (_i64, [_vp, _i64, _i32, _f32, _i32]),
    "flanhip_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "flanhip_free": (C.c_int, [_vp]),
    "flanhip_upload": (C.c_int, [_vp, _vp, C.c_size_t]),
    "flanhip_download": (C.c_int, [_vp, _vp, C.c_size_t]),
    "flanhip_touch_pages": (C.c_int, [_vp, C.c_size_t]),
    "flanhip_host_workers": (C.c_int, []),
    "flanhip_parallel_for": (C.c_int, [C.c_int, _vp, _vp]),
    "flanhip_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "flanhip_stream_destroy": (C.c_int, [_vp]),
    "flanhip_host_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "flanhip_host_free": (C.c_int, [_vp]),
    "flanhip_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "flanhip_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "flanhip_memset": (C.c_int, [_vp, _i32, C.c_size_t, _vp]),
    "flanhip_stream_synchronize": (C.c_int, [_vp]),
    "flanhip_wait_cancellable": (C.c_int, [_vp, _vp]),
    "flanhip_wait_cancellable_fn": (C.c_int, [_vp, _vp, _vp]),
    "flanhip_analyze": (C.c_int, [_vp, _i64, _i64, _f32, _i32, _i32, _i32, _vp, C.POINTER(_i64), _vp]),
    "flanhip_analyze_dev": (C.c_int, [_vp, _i64, _i64, _f32, _i32, _i32, _i32, _vp, _vp]),
    "flanhip_synthesize": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _i32, _vp, C.POINTER(_i32), _vp]),
    "flanhip_synthesize_workspace_bytes": (C.c_size_t, [_i64, _i64, _i32, _f32, _f32, _i32]),
    "flanhip_synthesize_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "flanhip_analyze_dev_fused": (C.c_int, [_vp, _i64, _i64, _f32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "flanhip_synthesize_dev_fused": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "flanhip_synthesize_dev_stages": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "flanhip_debug_option": (None, [_i32, _i32]),
    "flanhip_debug_kernel_scratch_bytes": (_i32, [_i32]),
    "flanhip_modify_time": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _vp, _i64, _vp, _vp]),
    "flanhip_modify_time_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _vp, _i64, _vp, _vp]),
    "flanhip_modify_time_dev_fused": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _vp, _i64, _vp, _i32, _vp, _vp]),
    "flanhip_synthesize_dev_fused_checked": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "flanhip_stretch_map_dev": (C.c_int, [_vp, _i64, _i32, _f32, _i32, _vp, _vp]),
    "flanhip_stretch_map_const_dev": (C.c_int, [_f32, _vp, _i64, _i32, _f32, _i32, _vp, _vp]),
    "flanhip_fill_dev": (C.c_int, [_vp, _i64, _f32, _vp]),
    "flanhip_modify_frequency": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "flanhip_modify_frequency_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "flanhip_modify_time_interp_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _vp, _i64, _i32, _vp, _vp]),
    "flanhip_modify_time_interp_dev_fused": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    "flanhip_modify_frequency_interp_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _i32, _vp, _vp]),
    "flanhip_repitch_interp_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _i32, _vp, _vp]),
    "flanhip_repitch_map_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp]),
    "flanhip_repitch_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp]),
    "flanhip_shape_affine": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "flanhip_shape_affine_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "flanhip_shape_table_dev": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _i32, _vp, _vp]),
    "flanhip_replace_amplitudes_dev": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _vp, _f32, _vp, _vp]),
    "flanhip_subtract_amplitudes_dev": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _vp, _f32, _vp, _vp]),
    "flanhip_resonate_out_frames": (_i64, [_i64, _f32, _f32, _i32]),
    "flanhip_resonate_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _i64, _vp, _f32, _vp, _vp]),
    "flanhip_n_loudest_partials_dev": (C.c_int, [_vp, _i64, _i64, _i32, _vp, C.c_int32, _i32, _vp, _vp]),
    "flanhip_desample_dev": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _f32, _i32, _vp, _vp]),
    "flanhip_interp_table_create": (C.c_int, [_vp, _vp]),
    "flanhip_interp_table_destroy": (C.c_int, [_i32]),
    "flanhip_time_extrapolate_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i64, _i64, _i64, _vp, _vp, _vp]),
    "flanhip_shape_affine_dev_fused": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _i32, _vp, _vp]),
    "flanhip_shape_table_dev_fused": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _f32, _vp, _i32, _vp, _vp]),
    "flanhip_get_frame_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _vp, _vp]),
    "flanhip_select_frames_dev": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp]),
    "flanhip_freeze_plan": (_i64, [_i64, _f32, _i32, _vp, _vp, _i32, _vp]),
    "flanhip_cut_frames_range": (C.c_int, [_i64, _i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "flanhip_cut_frames_dev": (C.c_int, [_vp, _i64, _i64, _i32, _i64, _i64, _vp, _vp]),
    "flanhip_place_frames_dev": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _i64, _vp]),
    "flanhip_select_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _vp, _i64, _vp, _vp]),
    "flanhip_harmonic_scale_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _i32, _i32, _vp, _vp]),
    "flanhip_modify_out_frames": (_i64, [_vp, _i64, _i32, _f32, _i32]),
    "flanhip_modify_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _vp, _vp, _i32, _i64, _vp, _vp]),
    "flanhip_stretch_spline_out_frames": (_i64, [_vp, _i64]),
    "flanhip_stretch_spline_dev": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp]),
    "flanhip_smear_time_plan": (C.c_int, [_i64, _i32, _f32, _i32, _vp, _f32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "flanhip_smear_time_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _i32, _vp, _f32, _vp, _i32, _vp, _i64, _i32, _i64, _vp, _vp]),
    "flanhip_mid_side_dev": (C.c_int, [_vp, _i64, _vp, _vp]),
    "flanhip_resample_out_frames": (_i64, [_i64, _f32, _f32]),
    "flanhip_resample": (C.c_int, [_vp, _i64, _i64, _f32, _f32, _vp, _vp]),
    "flanhip_resample_dev": (C.c_int, [_vp, _i64, _i64, _f32, _f32, _vp, _vp]),
    "flanhip_synthesize_prepass_dev": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp]),
    "flanhip_synthesize_dev_carry": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "flanhip_comm_unique_id": (C.c_int, [C.c_char_p]),
    "flanhip_comm_init": (C.c_int, [C.c_char_p, _i32, _i32, C.POINTER(_vp)]),
    "flanhip_comm_destroy": (C.c_int, [_vp]),
    "flanhip_allgather_audio": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "flanhip_noise_dev": (C.c_int, [_vp, _i64, _i64, C.c_uint32, _vp]),
    "flanhip_sqdiff_dev": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "flanhip_copy_dev": (C.c_int, [_vp, _vp, _i64, _vp]),
}

EXPORTS = sorted(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here == the library does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.flanhip_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        raise FlanHipError(rc, last_error())
    return rc


def _ptr(a):
    return a.ctypes.data_as(_vp)


# ---------------------------------------------------------------------------------------------------------------
# host-buffer wrappers (numpy in, numpy out) -- exactly the calls the C++ flan::Audio / flan::PV classes make
# ---------------------------------------------------------------------------------------------------------------

def analyze(audio, sample_rate, window=2048, hop=128, dft=4096):
    """Audio::convert_to_PV.  audio float32 [ch][n] -> float32 [ch][F][dft/2+1][2] (m, f)."""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    F = lib.flanhip_num_pv_frames(n, hop)
    out = np.empty((ch, F, dft // 2 + 1, 2), np.float32)
    got = _i64(0)
    check(lib.flanhip_analyze(_ptr(audio), ch, n, sample_rate, window, hop, dft, _ptr(out), C.byref(got), None))
    assert got.value == F
    return out


def synthesize(pv, sample_rate, analysis_rate, window):
    """PV::convert_to_audio.  pv float32 [ch][F][bins][2] -> (float32 [ch][F*hop], nan_flag)."""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    hop = lib.flanhip_hop_size(sample_rate, analysis_rate)
    out = np.empty((ch, F * hop), np.float32)
    flag = _i32(0)
    check(lib.flanhip_synthesize(_ptr(pv), ch, F, bins, sample_rate, analysis_rate, window, _ptr(out), C.byref(flag), None))
    return out, flag.value


def modify_time(pv, sample_rate, hop, mod_seconds):
    pv = np.ascontiguousarray(pv, np.float32)
    mod = np.ascontiguousarray(mod_seconds, np.float32)
    ch, F, bins, _ = pv.shape
    Fo = lib.flanhip_modify_time_out_frames(_ptr(mod), F, bins, sample_rate, hop)
    out = np.empty((ch, max(Fo, 0), bins, 2), np.float32)
    if Fo > 0:
        check(lib.flanhip_modify_time(_ptr(pv), ch, F, bins, sample_rate, hop, _ptr(mod), Fo, _ptr(out), None))
    return out


def modify_frequency(pv, sample_rate, mod_hz, in_modified):
    pv = np.ascontiguousarray(pv, np.float32)
    mod = np.ascontiguousarray(mod_hz, np.float32)
    inm = np.ascontiguousarray(in_modified, np.float32)
    ch, F, bins, _ = pv.shape
    out = np.empty_like(pv)
    check(lib.flanhip_modify_frequency(_ptr(pv), ch, F, bins, sample_rate, _ptr(mod), _ptr(inm), _ptr(out), None))
    return out


def resample(audio, src_rate, dst_rate):
    """Audio::resample (r8brain's single-step ratios and its block convolver + whole-stepping interpolator ratios, e.g. 44.1 <-> 48 kHz).
    audio float32 [ch][n] -> float32 [ch][n_out]"""
    audio = np.ascontiguousarray(audio, np.float32)
    ch, n = audio.shape
    n_out = lib.flanhip_resample_out_frames(n, src_rate, dst_rate)
    out = np.empty((ch, n_out), np.float32)
    check(lib.flanhip_resample(_ptr(audio), ch, n, src_rate, dst_rate, _ptr(out), None))
    return out


def shape_affine(pv, sample_rate, a, b, c, d, use_shift_alignment=False):
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    out = np.empty_like(pv)
    check(lib.flanhip_shape_affine(_ptr(pv), ch, F, bins, sample_rate, a, b, c, d, int(use_shift_alignment), _ptr(out), None))
    return out


class DeviceArray:
    """A device buffer owned through the C ABI (flanhip_malloc / flanhip_free), optionally filled from a numpy array."""

    def __init__(self, nbytes=None, host=None):
        if host is not None:
            host = np.ascontiguousarray(host)
            nbytes = host.nbytes
        self.nbytes = int(nbytes)
        p = _vp()
        check(lib.flanhip_malloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value
        if host is not None and self.nbytes:
            check(lib.flanhip_memcpy_h2d(_vp(self.ptr), _ptr(host), self.nbytes, None))

    def data_ptr(self):
        return self.ptr

    def to_host(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert out.nbytes == self.nbytes, (out.nbytes, self.nbytes)
        if self.nbytes:
            check(lib.flanhip_memcpy_d2h(_ptr(out), _vp(self.ptr), self.nbytes, None))
        check(lib.flanhip_stream_synchronize(None))
        return out

    def __del__(self):
        if getattr(self, "ptr", None):
            lib.flanhip_free(_vp(self.ptr))
            self.ptr = None


def _grid_or_const(grid):
    """(device pointer or None, constant) for a sampled user function: numpy grid, or a python scalar"""
    if np.isscalar(grid):
        return None, float(grid), None
    d = DeviceArray(host=np.ascontiguousarray(grid, np.float32))
    return _vp(d.ptr), 0.0, d


def _combine_amplitudes(fn, pv, src, amount):
    pv = np.ascontiguousarray(pv, np.float32)
    src = np.ascontiguousarray(src, np.float32)
    ch, F, bins, _ = pv.shape
    sch, sF, sbins, _ = src.shape
    d_pv, d_src, d_out = DeviceArray(host=pv), DeviceArray(host=src), DeviceArray(pv.nbytes)
    a_ptr, a_const, _keep = _grid_or_const(amount)
    check(fn(_vp(d_pv.ptr), ch, F, bins, _vp(d_src.ptr), sch, sF, sbins, a_ptr, a_const, _vp(d_out.ptr), None))
    return d_out.to_host(pv.shape)


def repitch(pv, sample_rate, factor_grid, interp=0):
    """PV::repitch.  factor_grid: float32 [F][bins] (the sampled factor); interp: FLANHIP_INTERP_*; returns the repitched PV"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    g = np.ascontiguousarray(factor_grid, np.float32)
    assert g.shape == (F, bins)
    d_pv, d_g, d_out = DeviceArray(host=pv), DeviceArray(host=g), DeviceArray(pv.nbytes)
    check(lib.flanhip_repitch_interp_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, _vp(d_g.ptr), interp, _vp(d_out.ptr), None))
    return d_out.to_host(pv.shape)


def modify_time_interp(pv, sample_rate, hop, mod_seconds, interp):
    """PV::modify_time with a named Interpolator (FLANHIP_INTERP_*), device entry point"""
    pv = np.ascontiguousarray(pv, np.float32)
    mod = np.ascontiguousarray(mod_seconds, np.float32)
    ch, F, bins, _ = pv.shape
    Fo = lib.flanhip_modify_time_out_frames(_ptr(mod), F, bins, sample_rate, hop)
    if Fo <= 0:
        return np.empty((ch, 0, bins, 2), np.float32)
    d_pv, d_mod, d_out = DeviceArray(host=pv), DeviceArray(host=mod), DeviceArray(ch * Fo * bins * 8)
    check(lib.flanhip_modify_time_interp_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, hop, _vp(d_mod.ptr), Fo, interp, _vp(d_out.ptr), None))
    return d_out.to_host((ch, Fo, bins, 2))


def modify_frequency_interp(pv, sample_rate, mod_hz, in_modified, interp):
    """PV::modify_frequency with a named Interpolator (FLANHIP_INTERP_*), device entry point"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    d_pv, d_mod, d_in = DeviceArray(host=pv), DeviceArray(host=np.ascontiguousarray(mod_hz, np.float32)), DeviceArray(host=np.ascontiguousarray(in_modified, np.float32))
    d_out = DeviceArray(pv.nbytes)
    check(lib.flanhip_modify_frequency_interp_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, _vp(d_mod.ptr), _vp(d_in.ptr), interp, _vp(d_out.ptr), None))
    return d_out.to_host(pv.shape)


def replace_amplitudes(pv, src, amount):
    """PV::replace_amplitudes.  amount: float32 [F][bins] or a scalar"""
    return _combine_amplitudes(lib.flanhip_replace_amplitudes_dev, pv, src, amount)


def subtract_amplitudes(pv, src, amount):
    """PV::subtract_amplitudes"""
    return _combine_amplitudes(lib.flanhip_subtract_amplitudes_dev, pv, src, amount)


def resonate(pv, sample_rate, hop, length_seconds, decay):
    """PV::resonate.  decay: float32 [Fo][bins] over the OUTPUT's domain, or a scalar"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    Fo = int(lib.flanhip_resonate_out_frames(F, length_seconds, sample_rate, hop))
    assert Fo >= F, Fo
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(ch * Fo * bins * 8)
    d_ptr, d_const, _keep = _grid_or_const(decay)
    check(lib.flanhip_resonate_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, hop, Fo, d_ptr, d_const, _vp(d_out.ptr), None))
    return d_out.to_host((ch, Fo, bins, 2))


def n_loudest_partials(pv, n, remove=False):
    """PV::retain_n_loudest_partials / remove_n_loudest_partials.  n: int32 [F] or a scalar"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(pv.nbytes)
    if np.isscalar(n):
        n_ptr, n_const, _keep = None, int(n), None
    else:
        _keep = DeviceArray(host=np.ascontiguousarray(n, np.int32))
        n_ptr, n_const = _vp(_keep.ptr), 0
    check(lib.flanhip_n_loudest_partials_dev(_vp(d_pv.ptr), ch, F, bins, n_ptr, n_const, int(remove), _vp(d_out.ptr), None))
    return d_out.to_host(pv.shape)


INTERP_TABLE_INTERVALS = 65536


class InterpTable:
    """an Interpolator built from a callable (Utility/Interpolator.h), sampled at i / 65536 and at NaN and registered with the library: use
    `.kind` wherever a FLANHIP_INTERP_* kind is taken; a context manager (the table is destroyed on exit)"""

    def __init__(self, fn):
        n = INTERP_TABLE_INTERVALS
        xs = np.arange(n + 1, dtype=np.float32) * np.float32(1.0 / n)
        samples = np.empty(n + 2, np.float32)
        samples[:n + 1] = [fn(np.float32(x)) for x in xs]
        samples[n + 1] = fn(np.float32(np.nan))
        kind = C.c_int(-1)
        check(lib.flanhip_interp_table_create(_ptr(samples), C.byref(kind)))
        self.kind = kind.value

    def close(self):
        if self.kind >= 0:
            check(lib.flanhip_interp_table_destroy(self.kind))
            self.kind = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def desample(pv, ratio, interp=0):
    """PV::desample.  ratio: float32 [F][bins] or a scalar; interp: FLANHIP_INTERP_*"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(pv.nbytes)
    r_ptr, r_const, _keep = _grid_or_const(ratio)
    check(lib.flanhip_desample_dev(_vp(d_pv.ptr), ch, F, bins, r_ptr, r_const, interp, _vp(d_out.ptr), None))
    return d_out.to_host(pv.shape)


def time_extrapolate(pv, sample_rate, start_frame, end_frame, out_frames, interp_samples):
    """PV::time_extrapolate after its input validation.  interp_samples: float32 [out_frames - start_frame]"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    samples = np.ascontiguousarray(interp_samples, np.float32)
    assert samples.shape == (out_frames - start_frame,)
    d_pv, d_s, d_out = DeviceArray(host=pv), DeviceArray(host=samples), DeviceArray(ch * out_frames * bins * 8)
    check(lib.flanhip_time_extrapolate_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, start_frame, end_frame, out_frames,
                                           _vp(d_s.ptr), _vp(d_out.ptr), None))
    return d_out.to_host((ch, out_frames, bins, 2))


def get_frame(pv, frame_pos, interp=0):
    """PV::get_frame at the (already clamped) fractional frame position.  Returns [ch][1][bins][2]"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(ch * bins * 8)
    check(lib.flanhip_get_frame_dev(_vp(d_pv.ptr), ch, F, bins, frame_pos, interp, _vp(d_out.ptr), None))
    return d_out.to_host((ch, 1, bins, 2))


def freeze_plan(num_frames, sample_rate, hop, times, lengths):
    """PV::freeze's timing logic: int32 [out_frames], the input frame of every output frame (-1: stays zero)"""
    times = np.ascontiguousarray(times, np.float32)
    lengths = np.ascontiguousarray(lengths, np.float32)
    assert times.shape == lengths.shape and times.ndim == 1
    n = len(times)
    tp = times.ctypes.data_as(_vp) if n else None
    lp = lengths.ctypes.data_as(_vp) if n else None
    Fo = lib.flanhip_freeze_plan(num_frames, sample_rate, hop, tp, lp, n, None)
    if Fo < 0:
        raise FlanHipError(-1, "flanhip_freeze_plan: bad arguments")
    src = np.empty(Fo, np.int32)
    lib.flanhip_freeze_plan(num_frames, sample_rate, hop, tp, lp, n, src.ctypes.data_as(_vp))
    return src


def select_frames(pv, src_frames):
    """out[c][o] = pv[c][src_frames[o]], zero where src_frames[o] < 0 (the copy loops of PV::freeze)"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    src = np.ascontiguousarray(src_frames, np.int32)
    Fo = len(src)
    d_pv, d_src, d_out = DeviceArray(host=pv), DeviceArray(host=src), DeviceArray(ch * Fo * bins * 8)
    check(lib.flanhip_select_frames_dev(_vp(d_pv.ptr), ch, F, bins, _vp(d_src.ptr), Fo, _vp(d_out.ptr), None))
    return d_out.to_host((ch, Fo, bins, 2))


def freeze(pv, sample_rate, hop, times, lengths):
    """PV::freeze"""
    return select_frames(pv, freeze_plan(np.shape(pv)[1], sample_rate, hop, times, lengths))


def cut_frames(pv, start, end):
    """PV::cut_frames; None for the null PV the reference returns"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    s, c = C.c_int32(0), C.c_int32(0)
    check(lib.flanhip_cut_frames_range(F, start, end, C.byref(s), C.byref(c)))
    if c.value <= 0:
        return None
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(ch * c.value * bins * 8)
    check(lib.flanhip_cut_frames_dev(_vp(d_pv.ptr), ch, F, bins, s.value, c.value, _vp(d_out.ptr), None))
    return d_out.to_host((ch, c.value, bins, 2))


def join(pvs):
    """PV::join: the format of the first input, the frames of all of them one after the other"""
    pvs = [np.ascontiguousarray(p, np.float32) for p in pvs]
    ch, _, bins, _ = pvs[0].shape
    Fo = sum(p.shape[1] for p in pvs)
    d_out = DeviceArray(ch * Fo * bins * 8)
    check(lib.flanhip_memset(_vp(d_out.ptr), 0, ch * Fo * bins * 8, None))
    at = 0
    for p in pvs:
        d_in = DeviceArray(host=p)
        check(lib.flanhip_place_frames_dev(_vp(d_in.ptr), p.shape[0], p.shape[1], p.shape[2], _vp(d_out.ptr), ch, Fo, bins, at, None))
        check(lib.flanhip_stream_synchronize(None))
        at += p.shape[1]
    return d_out.to_host((ch, Fo, bins, 2))


def select(pv, sample_rate, hop, selector_tf):
    """PV::select.  selector_tf: float32 [out_frames][bins][2] = the selector sampled over the output's domain"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    sel = np.ascontiguousarray(selector_tf, np.float32)
    Fo = sel.shape[0]
    assert sel.shape == (Fo, bins, 2)
    d_pv, d_sel, d_out = DeviceArray(host=pv), DeviceArray(host=sel), DeviceArray(ch * Fo * bins * 8)
    check(lib.flanhip_select_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, hop, _vp(d_sel.ptr), Fo, _vp(d_out.ptr), None))
    return d_out.to_host((ch, Fo, bins, 2))


def harmonic_scale(pv, sample_rate, series, mode):
    """PV::add_octaves (mode 0) / PV::add_harmonics (mode 1).  series: float32 [F][H]"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    series = np.ascontiguousarray(series, np.float32)
    assert series.ndim == 2 and series.shape[0] == F
    H = series.shape[1]
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(pv.nbytes)
    d_s = DeviceArray(host=series) if H else None
    check(lib.flanhip_harmonic_scale_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, _vp(d_s.ptr) if d_s else None, H, mode, _vp(d_out.ptr), None))
    return d_out.to_host(pv.shape)


def modify_out_frames(mod_tf, sample_rate, hop):
    mod = np.ascontiguousarray(mod_tf, np.float32)
    F, bins, _ = mod.shape
    return int(lib.flanhip_modify_out_frames(mod.ctypes.data_as(_vp), F, bins, sample_rate, hop))


def modify(pv, sample_rate, hop, mod_tf, in_f, interp=0, out_frames=None):
    """PV::modify.  mod_tf: float32 [F][bins][2] (seconds, Hz); in_f: float32 [ch][F][bins]"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    mod = np.ascontiguousarray(mod_tf, np.float32)
    in_f = np.ascontiguousarray(in_f, np.float32)
    assert mod.shape == (F, bins, 2) and in_f.shape == (ch, F, bins)
    Fo = modify_out_frames(mod, sample_rate, hop) if out_frames is None else out_frames
    if Fo <= 0:
        return None
    d_pv, d_mod, d_f, d_out = DeviceArray(host=pv), DeviceArray(host=mod), DeviceArray(host=in_f), DeviceArray(ch * Fo * bins * 8)
    check(lib.flanhip_modify_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, hop, _vp(d_mod.ptr), _vp(d_f.ptr), interp, Fo, _vp(d_out.ptr), None))
    return d_out.to_host((ch, Fo, bins, 2))


def stretch_spline(pv, steps):
    """PV::stretch_spline.  steps: uint32 [F-1], every one >= 1"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    steps = np.ascontiguousarray(steps, np.uint32)
    assert steps.shape == (F - 1,)
    Fo = int(lib.flanhip_stretch_spline_out_frames(steps.ctypes.data_as(_vp), F))
    if Fo < 0:
        raise FlanHipError(-1, "flanhip_stretch_spline_out_frames: bad steps")
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(ch * Fo * bins * 8)
    check(lib.flanhip_stretch_spline_dev(_vp(d_pv.ptr), ch, F, bins, steps.ctypes.data_as(_vp), Fo, _vp(d_out.ptr), None))
    check(lib.flanhip_stream_synchronize(None))
    return d_out.to_host((ch, Fo, bins, 2))


def smear_time_plan(num_frames, num_bins, sample_rate, hop, smear):
    """(true_left, out_frames, dist_samples_2) of PV::smear_time for a smear grid float32 [F][bins] or a scalar"""
    left, Fo, half = C.c_int32(0), C.c_int64(0), C.c_int32(0)
    if np.isscalar(smear):
        check(lib.flanhip_smear_time_plan(num_frames, num_bins, sample_rate, hop, None, float(smear), C.byref(left), C.byref(Fo), C.byref(half)))
    else:
        g = np.ascontiguousarray(smear, np.float32)
        assert g.shape == (num_frames, num_bins)
        check(lib.flanhip_smear_time_plan(num_frames, num_bins, sample_rate, hop, g.ctypes.data_as(_vp), 0.0, C.byref(left), C.byref(Fo), C.byref(half)))
    return left.value, Fo.value, half.value


def smear_time(pv, sample_rate, hop, smear, granularity, dist, true_left, out_frames):
    """PV::smear_time.  smear: float32 [F][bins] or scalar; granularity: int32 [F][bins] or scalar; dist: float32 [2 * dist_samples_2]"""
    pv = np.ascontiguousarray(pv, np.float32)
    ch, F, bins, _ = pv.shape
    s_ptr, s_const, _k1 = _grid_or_const(smear)
    if np.isscalar(granularity):
        g_ptr, g_const, _k2 = None, int(granularity), None
    else:
        _k2 = DeviceArray(host=np.ascontiguousarray(granularity, np.int32))
        g_ptr, g_const = _vp(_k2.ptr), 0
    dist = np.ascontiguousarray(dist, np.float32)
    d_dist = DeviceArray(host=dist) if len(dist) else None
    d_pv, d_out = DeviceArray(host=pv), DeviceArray(ch * out_frames * bins * 8)
    check(lib.flanhip_smear_time_dev(_vp(d_pv.ptr), ch, F, bins, sample_rate, hop, s_ptr, s_const, g_ptr, g_const,
                                     _vp(d_dist.ptr) if d_dist else None, len(dist), true_left, out_frames, _vp(d_out.ptr), None))
    return d_out.to_host((ch, out_frames, bins, 2))


# ---------------------------------------------------------------------------------------------------------------
# device-pointer wrappers: arguments are objects with .data_ptr() (torch tensors on the GPU) or raw ints
# ---------------------------------------------------------------------------------------------------------------

def _dp(t):
    if t is None:
        return None
    return _vp(t.data_ptr() if hasattr(t, "data_ptr") else int(t))


def analyze_dev(d_audio, ch, n, sample_rate, window, hop, dft, d_out, stream=None):
    check(lib.flanhip_analyze_dev(_dp(d_audio), ch, n, sample_rate, window, hop, dft, _dp(d_out), _vp(stream or 0)))


def analyze_dev_fused(d_audio, ch, n, sample_rate, window, hop, dft, d_out, d_ws, stream=None):
    check(lib.flanhip_analyze_dev_fused(_dp(d_audio), ch, n, sample_rate, window, hop, dft, _dp(d_out), _dp(d_ws), _vp(stream or 0)))


def synthesize_dev_fused(d_pv, ch, F, bins, sample_rate, analysis_rate, window, d_out, d_ws, d_nan=None, stream=None):
    check(lib.flanhip_synthesize_dev_fused(_dp(d_pv), ch, F, bins, sample_rate, analysis_rate, window, _dp(d_out), _dp(d_ws),
                                           _dp(d_nan), _vp(stream or 0)))


def synthesize_dev_stages(d_pv, ch, F, bins, sample_rate, analysis_rate, window, d_out, d_ws, d_nan, presummed, stages, stream=None):
    """flanhip_synthesize_dev (presummed 0) / _fused (1) / _fused_checked (2) with the kernels to launch as a per-call argument
    (1 k_phase_sums, 2 k_phase_scan, 4 k_synthesize, 8 k_ola_fixup): bench.py times one kernel at a time with it"""
    check(lib.flanhip_synthesize_dev_stages(_dp(d_pv), ch, F, bins, sample_rate, analysis_rate, window, _dp(d_out), _dp(d_ws),
                                            _dp(d_nan), presummed, stages, _vp(stream or 0)))


# flanhip_debug_option: per-thread test / A-B hooks (include/flanhip.h)
DEBUG_CHAIN_LEN, DEBUG_TARGET_CHAINS, DEBUG_FORCE_GENERIC, DEBUG_NO_FAST_DIV = 0, 1, 2, 3
DEBUG_ANA_VARIANT, DEBUG_SYN_VARIANT, DEBUG_ANA4096_OLD, DEBUG_SYN4096_OLD, DEBUG_RESAMPLE_DIRECT, DEBUG_FORCE_DIRECT, DEBUG_INLINE_FIXUP, DEBUG_WIDE_OFFSETS = 4, 5, 6, 7, 8, 9, 10, 11
_DEBUG_NAMES = {"chain_len": 0, "target_chains": 1, "force_generic": 2, "no_fast_div": 3, "ana_variant": 4, "syn_variant": 5,
                "ana4096_old": 6, "syn4096_old": 7, "resample_direct": 8, "force_direct": 9, "inline_fixup": 10, "wide_offsets": 11, "no_sub": 12}


class debug_options:
    """with fa.debug_options(chain_len=37): ...   -- the calling thread's hooks set for the block, cleared (0) afterwards"""

    def __init__(self, **kw):
        self.kw = {_DEBUG_NAMES[k]: int(v) for k, v in kw.items()}

    def __enter__(self):
        for k, v in self.kw.items():
            lib.flanhip_debug_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            lib.flanhip_debug_option(k, 0)
        return False


def synthesize_workspace_bytes(ch, F, bins, sample_rate, analysis_rate, window):
    return int(lib.flanhip_synthesize_workspace_bytes(ch, F, bins, sample_rate, analysis_rate, window))


def synthesize_dev(d_pv, ch, F, bins, sample_rate, analysis_rate, window, d_out, d_ws, d_nan=None, stream=None):
    check(lib.flanhip_synthesize_dev(_dp(d_pv), ch, F, bins, sample_rate, analysis_rate, window, _dp(d_out), _dp(d_ws),
                                     _dp(d_nan), _vp(stream or 0)))
