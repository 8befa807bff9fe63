"""The C-ABI library loads and exports every symbol include/flanhip.h declares; the pure-host helpers work and the
compute entry points fail loudly (FLANHIP_ERR_NO_DEVICE) when no GPU is visible.  No compute here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "flanhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flanhip_\w+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "flan_amd", "libflanhip.so"))
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    import flan_amd
    assert sorted(flan_amd.EXPORTS) == declared_symbols()


def test_shape_helpers():
    import flan_amd
    lib = flan_amd.lib
    assert lib.flanhip_version() >= 100
    assert lib.flanhip_num_pv_frames(240000, 512) == 469           # AudioPV.cpp:17 integer division + 1
    assert lib.flanhip_num_pv_frames(100, 512) == 1
    assert lib.flanhip_num_pv_frames(0, 512) == 1
    assert lib.flanhip_num_pv_frames(10, 0) == -1
    assert lib.flanhip_hop_size(48000.0, 93.75) == 512
    assert lib.flanhip_hop_size(48000.0, np.float32(48000.0) / np.float32(333)) in (332, 333)   # float truncation quirk, PVBuffer.cpp:381-384
    mod = np.array([[0.0, 0.5], [1.0, 0.25]], np.float32)
    # ceil( 1.0 s * 48000 / 512 ) = 94
    assert lib.flanhip_modify_time_out_frames(mod.ctypes.data_as(ctypes.c_void_p), 2, 2, 48000.0, 512) == 94
    assert lib.flanhip_synthesize_workspace_bytes(2, 100, 1025, 48000.0, 93.75, 2048) > 0
    assert lib.flanhip_synthesize_workspace_bytes(2, 100, 1001, 48000.0, 93.75, 2048) == 0       # dft 2000: unsupported


def test_no_device_is_loud_not_a_fallback():
    import torch
    import flan_amd
    if torch.cuda.is_available():
        pytest.skip("GPU visible")
    assert flan_amd.lib.flanhip_device_count() == 0
    x = np.zeros((1, 4096), np.float32)
    with pytest.raises(flan_amd.FlanHipError) as e:
        flan_amd.analyze(x, 48000.0, 2048, 512, 2048)
    assert e.value.code == flan_amd.ERR_NO_DEVICE
    pv = np.zeros((1, 9, 1025, 2), np.float32)
    with pytest.raises(flan_amd.FlanHipError) as e:
        flan_amd.synthesize(pv, 48000.0, 93.75, 2048)
    assert e.value.code == flan_amd.ERR_NO_DEVICE
    with pytest.raises(flan_amd.FlanHipError) as e:
        flan_amd.shape_affine(pv, 48000.0, 1, 0, 1, 0)
    assert e.value.code == flan_amd.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """flan_amd/, include/ and the host library must not reference oracle/ (checker != product)"""
    bad = []
    for base in ("flan_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"import\s+oracle|from\s+oracle|oracle_lib|liboracle|flan_oracle|-loracle|oracle/_ref|libflanref", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_worker_pool_and_page_touch_without_a_device():
    """flanhip_parallel_for / flanhip_touch_pages are host plumbing: they work with no GPU (the C++ classes sample callables on them)"""
    import ctypes as C
    import numpy as np
    import flan_amd
    lib = flan_amd.lib
    assert lib.flanhip_host_workers() >= 1
    hits = np.zeros(10007, np.int32)
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)

    def body(ctx, i):
        hits[i] += 1                                  # every index is visited by exactly one thread
    cb = CB(body)
    assert lib.flanhip_parallel_for(len(hits), C.cast(cb, C.c_void_p), None) == 0
    assert np.all(hits == 1)
    assert lib.flanhip_parallel_for(0, C.cast(cb, C.c_void_p), None) == 0
    assert lib.flanhip_parallel_for(5, None, None) != 0        # a null task is an error, not a crash
    buf = np.full(8 << 20, 7, np.uint8)
    assert lib.flanhip_touch_pages(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    assert buf[::4096].sum() == 0 and buf[1] == 7               # one zero per page, nothing else touched
