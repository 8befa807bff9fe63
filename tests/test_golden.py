"""Golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py): the oracle must reproduce them bit for bit
on any machine (CPU test), and the HIP path must match them within the parity tolerances (GPU test)."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
SR = 48000.0


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    ch, n, W, hop, dft = [int(v) for v in g["params"]]
    pv = O.analyze(g["audio"], SR, W, hop, dft)
    assert np.array_equal(_bits(pv), _bits(g["pv"]))
    out, _ = O.synthesize(g["pv"], SR, np.float32(SR) / np.float32(hop), W)
    assert np.array_equal(_bits(out), _bits(g["out"]))
    if "stretch2" in g:
        F, bins = pv.shape[1], pv.shape[2]
        two = np.full((F, bins), 2.0, np.float32)
        assert np.array_equal(_bits(O.stretch(g["pv"], SR, hop, two)), _bits(g["stretch2"]))
        assert np.array_equal(_bits(O.repitch(g["pv"], SR, two)), _bits(g["repitch2"]))
        assert np.array_equal(_bits(O.shape_affine(g["pv"], SR, 1.0, 0.0, 1.0, 100.0, False)), _bits(g["shape_f_plus_100"]))
        assert np.array_equal(_bits(O.shape_affine(g["pv"], SR, 1.0, 0.0, 2.0, 0.0, True)), _bits(g["shape_f_times_2_aligned"]))


def test_there_are_fixtures():
    assert len(FILES) >= 6


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_matches_golden(path):
    import flan_amd as fa
    g = np.load(path)
    ch, n, W, hop, dft = [int(v) for v in g["params"]]
    pv = fa.analyze(g["audio"], SR, W, hop, dft)
    m_g, m_r = pv[..., 0].astype(np.float64), g["pv"][..., 0].astype(np.float64)
    if np.any(m_r):
        assert np.sqrt(np.sum((m_g - m_r) ** 2) / np.sum(m_r ** 2)) <= 1e-5
    else:
        assert not np.any(m_g)
    out, flag = fa.synthesize(g["pv"], SR, np.float32(SR) / np.float32(hop), W)
    assert flag == 0
    assert np.sqrt(np.mean((out.astype(np.float64) - g["out"].astype(np.float64)) ** 2)) <= 1e-5
    if "stretch2" in g:
        F, bins = pv.shape[1], pv.shape[2]
        two = np.full((F, bins), 2.0, np.float32)
        got = fa.modify_time(g["pv"], SR, hop, O.stretch_map(two, SR, hop))
        assert np.array_equal(_bits(got), _bits(g["stretch2"]))
        mod_hz, inmod = O.repitch_map(g["pv"], SR, two)
        assert np.array_equal(_bits(fa.modify_frequency(g["pv"], SR, mod_hz, inmod)), _bits(g["repitch2"]))
        assert np.array_equal(_bits(fa.shape_affine(g["pv"], SR, 1.0, 0.0, 1.0, 100.0, False)), _bits(g["shape_f_plus_100"]))
        assert np.array_equal(_bits(fa.shape_affine(g["pv"], SR, 1.0, 0.0, 2.0, 0.0, True)), _bits(g["shape_f_times_2_aligned"]))
