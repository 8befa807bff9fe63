""".flan PV files: the host library's PVBuffer::save / load against the reference's own PVBuffer.cpp,
compiled unmodified into oracle/_ref/libflanref.so.  Files written by either side must be byte-identical and load to
bit-identical data, including the reference's header asymmetry (save writes the hop where load reads analysis_rate)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = O.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libflanref.so not built (no /root/reference)")


@pytest.fixture(scope="module")
def host():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "flan_amd", "host")], check=True)
    C.CDLL(os.path.join(ROOT, "flan_amd", "libflanhip.so"), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(os.path.join(ROOT, "flan_amd", "libflan_host.so"))
    lib.flan_pv_save_file.argtypes = [O.RefPVFormat, O.f32p, C.c_char_p]
    lib.flan_pv_load_file.restype = C.c_int64
    lib.flan_pv_load_file.argtypes = [C.c_char_p, C.POINTER(O.RefPVFormat), C.c_void_p, C.c_int64]
    return lib


def test_save_and_load_match_the_reference(host, tmp_path):
    x = O.noise(2, 6000, seed=4)
    sr, hop, dft, W = 48000.0, 256, 1024, 1024
    pv = O.analyze(x, sr, W, hop, dft)
    pv[0, 1, 5] = (3000.0, -100.0)            # beyond the +-1 clamp of m / dft, negative frequency
    ch, F, bins, _ = pv.shape
    fmt = O.RefPVFormat(ch, F, bins, sr, np.float32(sr) / np.float32(hop), W)
    a, b = str(tmp_path / "ref.flan").encode(), str(tmp_path / "ours.flan").encode()
    assert ref.ref_pv_save(fmt, pv.reshape(-1), a) == 1
    assert host.flan_pv_save_file(fmt, pv.reshape(-1), b) == 1
    bytes_a, bytes_b = open(a, "rb").read(), open(b, "rb").read()
    assert len(bytes_a) == len(bytes_b) == 12 + 8 + 30 + 8 + pv.size * 3
    # byte 10/11 of the RIFF type tag come from reading past the literal "PV" in the reference (Bytes.cpp writeBytes of a
    # 3-byte string literal); everything else must agree
    assert bytes_a[:10] == bytes_b[:10] and bytes_a[12:] == bytes_b[12:]

    for path in (a, b):
        f_r, f_o = O.RefPVFormat(), O.RefPVFormat()
        mf_r = np.zeros(pv.size, np.float32); mf_o = np.zeros(pv.size, np.float32)
        assert ref.ref_pv_load(path, C.byref(f_r), mf_r.ctypes.data_as(C.c_void_p), pv.size // 2) == 1
        assert host.flan_pv_load_file(path, C.byref(f_o), mf_o.ctypes.data_as(C.c_void_p), pv.size // 2) == pv.size // 2
        for field, _ in O.RefPVFormat._fields_:
            assert getattr(f_r, field) == getattr(f_o, field), field
        assert f_o.analysis_rate == hop        # the asymmetry: the file holds the hop in that slot (PVBuffer.cpp:134 vs :245)
        assert np.array_equal(mf_r.view(np.uint32), mf_o.view(np.uint32))
        # 24-bit quantisation: m / dft and f / sample_rate
        back = mf_o.reshape(pv.shape)
        assert np.abs(back[..., 1] - np.clip(pv[..., 1], -sr, sr)).max() <= 2 * sr / 2 ** 23   # truncation to 24 bits + fp32 rounding
