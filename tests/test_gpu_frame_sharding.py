"""Frame-range sharding (secondary axis: few channels, long signals) emulated on the one GPU of the test box: the
ranks of a world of 2 / 3 / 5 run one after the other through exactly the calls a multi-GPU job makes
(flan_amd/sharding.py geometry, flanhip_analyze, flanhip_synthesize_prepass_dev, flanhip_synthesize_dev_carry) and their pieces
are put together on the host.  Analysis pieces must equal the unsharded PV bit for bit; the audio may differ from the unsharded
run only by the re-association of overlap sums at range / chain boundaries."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SR = 48000.0


def sharded_round_trip(fa, x, W, hop, dft, world):
    from flan_amd import sharding as S
    lib = fa.lib
    ch, n = x.shape
    bins = dft // 2 + 1
    F = n // hop + 1
    ar = np.float32(SR) / np.float32(hop)
    pad = S.pad_frames(W, hop)
    P = lambda d: C.c_void_p(d.ptr)
    ranges = S.frame_ranges(F, world)
    pieces, state, totals = [], [], []
    for fb, fe in ranges:                                                   # ---- every rank: analysis of its frame range, pre-pass
        s0, s1, j0 = S.analysis_slice(n, hop, W, fb, fe, F)
        local = fa.analyze(np.ascontiguousarray(x[:, s0:s1]), SR, W, hop, dft)
        piece = np.ascontiguousarray(local[:, j0:j0 + (fe - fb)])
        pieces.append(piece)
        rows = fe - fb + 2 * pad
        padded = np.zeros((ch, rows, bins, 2), np.float32)
        padded[:, pad:pad + fe - fb] = piece
        d_pv = fa.DeviceArray(host=padded)
        d_ws = fa.DeviceArray(fa.synthesize_workspace_bytes(ch, rows, bins, SR, float(ar), W))
        d_tot = fa.DeviceArray(ch * bins * 8)
        d_flag = fa.DeviceArray(host=np.zeros(1, np.int32))
        fa.check(lib.flanhip_synthesize_prepass_dev(P(d_pv), ch, rows, bins, SR, ar, W, P(d_ws), P(d_tot), P(d_flag), None))
        totals.append(d_tot.to_host((ch, bins), np.float64))
        state.append((d_pv, d_ws, d_flag, rows))
    carries = S.fold_carry(totals)                                          # ---- the one exchange: per-rank totals -> carries
    out = np.zeros((ch, F * hop), np.float32)
    for (fb, fe), (d_pv, d_ws, d_flag, rows), carry in zip(ranges, state, carries):   # ---- every rank: synthesis from its carry
        d_carry = fa.DeviceArray(host=np.ascontiguousarray(carry, np.float64))
        d_out = fa.DeviceArray(ch * rows * hop * 4)
        fa.check(lib.flanhip_synthesize_dev_carry(P(d_pv), ch, rows, bins, SR, ar, W, P(d_out), P(d_ws), P(d_carry), P(d_flag), None))
        S.place_local_output(out, d_out.to_host((ch, rows * hop)), fb, hop, pad)
        assert int(d_flag.to_host((1,), np.int32)[0]) == 0
    return np.concatenate(pieces, axis=1), out


@pytest.mark.parametrize("W,hop,dft,n", [(2048, 512, 2048, 150000), (1024, 256, 1024, 60001), (2048, 512, 4096, 90000), (2048, 128, 4096, 30000)])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_frame_sharded_round_trip(W, hop, dft, n, world):
    import flan_amd as fa
    x = O.noise(2, n, seed=world * 1000 + dft)
    pv_full = fa.analyze(x, SR, W, hop, dft)
    out_full, _ = fa.synthesize(pv_full, SR, np.float32(SR) / np.float32(hop), W)
    pv_sh, out_sh = sharded_round_trip(fa, x, W, hop, dft, world)
    assert pv_sh.shape == pv_full.shape
    assert np.array_equal(pv_sh.view(np.uint32), pv_full.view(np.uint32))            # analysis: bit for bit
    d = np.abs(out_sh.astype(np.float64) - out_full.astype(np.float64))
    print("\n[frame shards world=%d dft=%d hop=%d] audio max |d| = %.2e, rms = %.2e" % (world, dft, hop, d.max(), np.sqrt(np.mean(d ** 2))))
    assert d.max() <= 5e-6


def test_carry_is_the_phase_so_far():
    """the second half of a PV synthesised from the first half's total equals the second half of the whole synthesis"""
    import flan_amd as fa
    from flan_amd import sharding as S
    W, hop, dft = 2048, 512, 2048
    x = O.noise(1, 80000, seed=3)
    pv = fa.analyze(x, SR, W, hop, dft)
    _, out_sh = sharded_round_trip(fa, x, W, hop, dft, 2)
    out_full, _ = fa.synthesize(pv, SR, np.float32(SR) / np.float32(hop), W)
    half = (pv.shape[1] // 2 + 4) * hop
    assert np.abs(out_sh[:, half:] - out_full[:, half:]).max() <= 5e-6          # would be O(1) with a wrong carry
