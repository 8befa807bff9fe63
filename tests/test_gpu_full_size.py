"""BASELINE.json's full sizes on one GPU (config 2: stereo 60 s; config 4's per-GPU shard: 8 ch x 600 s, a 3.7 GB PV), checked
through properties that do not need a full-size CPU run:
  * head parity: the first frames of the long analysis / the first samples of the long synthesis depend only on the head of
    the input, so they are compared with the oracle run on that head (P1 / P2 tolerances);
  * chain invariance: cutting the frames into chains differently must not change the PV at all and the audio by more than
    re-association of the overlap sums;
  * shard invariance: one call over all channels == one call per channel, bit for bit (what multi-GPU sharding relies on);
  * the round trip reproduces its input (the reference's own gain 1.00074 at window 2048 / hop 512, SURVEY 8c anchor) within
    the reference algorithm's own phase drift."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

SR, W, HOP, DFT = 48000.0, 2048, 512, 2048
BINS = DFT // 2 + 1
AR = SR / HOP


def _p(t):
    return C.c_void_p(t.data_ptr())


def _sqdiff(fa, torch, a, b):
    r = torch.zeros(2, dtype=torch.float64, device=a.device)
    fa.check(fa.lib.flanhip_sqdiff_dev(_p(a), _p(b), a.numel(), _p(r), None))
    torch.cuda.synchronize()
    return [float(v) for v in r.cpu()]


@pytest.mark.parametrize("name,ch,seconds", [("config2 stereo 60 s", 2, 60), ("config4 shard 8 ch x 600 s", 8, 600)])
def test_full_size_properties(name, ch, seconds, monkeypatch):
    import torch
    import flan_amd as fa
    lib = fa.lib
    dev = torch.device("cuda", 0)
    n = seconds * 48000
    F = int(lib.flanhip_num_pv_frames(n, HOP))
    x = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(_p(x), ch, n, 4321, None))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, AR, W), dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fa.analyze_dev(x, ch, n, SR, W, HOP, DFT, pv)
    fa.synthesize_dev(pv, ch, F, BINS, SR, AR, W, out, ws, flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    assert bool(torch.isfinite(out).all())

    # ---- head parity against the oracle on the first K frames (channel 0 and the last channel)
    K = 200
    n_head = (K + 4) * HOP
    for c in (0, ch - 1):
        x_head = x[c:c + 1, :n_head].cpu().numpy()
        assert np.array_equal(x_head, O.noise(ch, n, 4321)[c:c + 1, :n_head]) if n <= 3_000_000 else True
        pv_ref = O.analyze(x_head, SR, W, HOP, DFT)[:, :K]
        pv_got = pv[c:c + 1, :K].cpu().numpy()
        m_r, m_g = pv_ref[..., 0].astype(np.float64), pv_got[..., 0].astype(np.float64)
        rel_m = np.sqrt(np.sum((m_g - m_r) ** 2) / np.sum(m_r ** 2))
        same_f = np.mean(pv_ref[..., 1].view(np.uint32) == pv_got[..., 1].view(np.uint32))
        out_ref, _ = O.synthesize(pv_got, SR, np.float32(SR) / np.float32(HOP), W)       # identical PV in (P2)
        valid = (K - 4) * HOP                                                            # samples no later frame reaches
        rms = np.sqrt(np.mean((out[c, :valid].cpu().numpy().astype(np.float64) - out_ref[0, :valid]) ** 2))
        print("\n[%s ch %d] head P1 rel_m=%.2e  f bit-identical=%.4f   head P2 rms=%.2e" % (name, c, rel_m, same_f, rms))
        assert rel_m <= 1e-5 and same_f >= 0.97 and rms <= 1e-5

    # ---- the round trip reproduces its input with the reference's gain 1.00074.  The reference algorithm itself
    # drifts: frequencies are stored in fp32, so the resynthesised phases walk away from the input's by ~6e-4 relative l2 per
    # second of signal (measured on the oracle: 6.4e-4 @1 s, 3.2e-3 @5 s, 1.25e-2 @20 s) -- checked over the first 10 seconds
    g = torch.tensor(1.00074, dtype=torch.float32, device=dev)
    lo, hi = W, 10 * 48000
    err = torch.linalg.vector_norm((out[:, lo:hi] - g * x[:, lo:hi]).double()) / torch.linalg.vector_norm(x[:, lo:hi].double())
    print("[%s] round trip vs 1.00074 x input over the first 10 s: relative l2 = %.2e" % (name, float(err)))
    assert float(err) <= 1.2e-2

    # ---- chain invariance (the chain-length hook of flanhip_debug_option, per thread)
    pv2 = torch.empty_like(pv)
    out2 = torch.empty_like(out)
    fa.lib.flanhip_debug_option(fa.DEBUG_CHAIN_LEN, 37)
    # the workspace layout follows the chain length: size it under the same setting
    ws37 = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, AR, W), dtype=torch.uint8, device=dev)
    fa.analyze_dev(x, ch, n, SR, W, HOP, DFT, pv2)
    fa.synthesize_dev(pv2, ch, F, BINS, SR, AR, W, out2, ws37, flag)
    fa.lib.flanhip_debug_option(fa.DEBUG_CHAIN_LEN, 0)
    torch.cuda.synchronize()
    assert bool(torch.equal(pv.view(torch.int32), pv2.view(torch.int32)))
    dmax = float((out - out2).abs().max())
    print("[%s] chain length 37 vs default: PV bit-identical, audio max |d| = %.2e" % (name, dmax))
    assert dmax <= 5e-6

    # ---- shard invariance: per-channel calls == the all-channel call, bit for bit
    for c in (0, ch - 1):
        pv_c = torch.empty((1, F, BINS, 2), dtype=torch.float32, device=dev)
        out_c = torch.empty((1, F * HOP), dtype=torch.float32, device=dev)
        fa.analyze_dev(x[c:c + 1], 1, n, SR, W, HOP, DFT, pv_c)
        torch.cuda.synchronize()
        assert bool(torch.equal(pv_c.view(torch.int32), pv[c:c + 1].view(torch.int32)))
        # synthesis: the chain length depends on the channel count, so pin it for an exact comparison
        fa.lib.flanhip_debug_option(fa.DEBUG_CHAIN_LEN, 64)
        ws_c = torch.empty(fa.synthesize_workspace_bytes(1, F, BINS, SR, AR, W), dtype=torch.uint8, device=dev)
        ws64 = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, AR, W), dtype=torch.uint8, device=dev)
        fa.synthesize_dev(pv_c, 1, F, BINS, SR, AR, W, out_c, ws_c, flag)
        fa.synthesize_dev(pv, ch, F, BINS, SR, AR, W, out2, ws64, flag)
        fa.lib.flanhip_debug_option(fa.DEBUG_CHAIN_LEN, 0)
        torch.cuda.synchronize()
        assert bool(torch.equal(out_c.view(torch.int32), out2[c:c + 1].view(torch.int32)))


@pytest.mark.parametrize("w,hop,dft,ch,seconds", [(512, 128, 512, 8, 600), (256, 64, 256, 8, 300), (8192, 2048, 8192, 8, 600), (4096, 1024, 16384, 8, 300),
                                                  # ... at fractions of a step (the API's default ratio and its default hop at the larger sizes) and dft 128
                                                  (4096, 256, 8192, 8, 120), (2048, 128, 16384, 4, 60), (128, 32, 128, 8, 300)])
def test_round6_kernel_families_at_full_size(w, hop, dft, ch, seconds):
    """The kernels of round 6 (pv_kernels_sub.h: dft 512 / 256, several chains per wavefront; pv_kernels_team.h: dft 8192 / 16384, teams of wavefronts) on
    long inputs (PVs of 3.7 - 7.4 GB, chains of hundreds of frames in several rounds), through size-independent properties: head parity against the oracle,
    the PV bit for bit and the audio to re-association under another chain cut, per-channel calls == the all-channel call, and the analysis of the signal's
    last 20 s as a signal of its own == the long run's last rows (the far end of every 32-bit offset)."""
    import torch
    import flan_amd as fa
    lib = fa.lib
    dev = torch.device("cuda", 0)
    bins, ar = dft // 2 + 1, SR / hop
    n = seconds * 48000
    F = int(lib.flanhip_num_pv_frames(n, hop))
    x = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(_p(x), ch, n, 777, None))
    pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
    out = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, ar, w), dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fa.analyze_dev(x, ch, n, SR, w, hop, dft, pv)
    fa.synthesize_dev(pv, ch, F, bins, SR, ar, w, out, ws, flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and bool(torch.isfinite(out).all())
    # ---- head parity (channel 0 and the last one)
    K = 120
    n_head = (K + 2 * (w // hop)) * hop
    for c in (0, ch - 1):
        x_head = x[c:c + 1, :n_head].cpu().numpy()
        pv_ref = O.analyze(x_head, SR, w, hop, dft)[:, :K]
        pv_got = pv[c:c + 1, :K].cpu().numpy()
        m_r, m_g = pv_ref[..., 0].astype(np.float64), pv_got[..., 0].astype(np.float64)
        rel_m = np.sqrt(np.sum((m_g - m_r) ** 2) / np.sum(m_r ** 2))
        same_f = np.mean(pv_ref[..., 1].view(np.uint32) == pv_got[..., 1].view(np.uint32))
        out_ref, _ = O.synthesize(pv_got, SR, np.float32(SR) / np.float32(hop), w)       # identical PV in (P2)
        valid = (K - w // hop) * hop
        rms = np.sqrt(np.mean((out[c, :valid].cpu().numpy().astype(np.float64) - out_ref[0, :valid]) ** 2))
        print("\n[(%d, %d, %d) %d ch x %d s, ch %d] head P1 rel_m=%.2e  f bit-identical=%.4f   head P2 rms=%.2e" % (w, hop, dft, ch, seconds, c, rel_m, same_f, rms))
        assert rel_m <= 1e-5 and same_f >= (0.95 if dft >= 256 else 0.90) and rms <= 1e-5      # (dft 128: 0.935 from every kernel generation, tests/test_gpu_conversions.py)
    # ---- another chain cut
    pv2 = torch.empty_like(pv)
    out2 = torch.empty_like(out)
    with fa.debug_options(chain_len=53):
        ws53 = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, SR, ar, w), dtype=torch.uint8, device=dev)
        fa.analyze_dev(x, ch, n, SR, w, hop, dft, pv2)
        fa.synthesize_dev(pv2, ch, F, bins, SR, ar, w, out2, ws53, flag)
    torch.cuda.synchronize()
    assert bool(torch.equal(pv.view(torch.int32), pv2.view(torch.int32)))
    dmax = float((out - out2).abs().max())
    print("[(%d, %d, %d)] chain length 53 vs default: PV bit-identical, audio max |d| = %.2e" % (w, hop, dft, dmax))
    assert dmax <= 5e-6
    del pv2, out2, ws53
    # ---- one channel alone == its rows of the all-channel call
    for c in (0, ch - 1):
        pv_c = torch.empty((1, F, bins, 2), dtype=torch.float32, device=dev)
        fa.analyze_dev(x[c:c + 1], 1, n, SR, w, hop, dft, pv_c)
        torch.cuda.synchronize()
        assert bool(torch.equal(pv_c.view(torch.int32), pv[c:c + 1].view(torch.int32)))
        del pv_c
    # ---- the last 20 s as a signal of its own: its rows (past the frames whose windows reach back over its start) are the long run's last rows
    tail_frames = (20 * 48000) // hop
    start = (n // hop - tail_frames) * hop                     # (a multiple of the hop: the tail's frame t' is the long run's frame start / hop + t')
    xt = x[:, start:].contiguous()
    Ft = int(lib.flanhip_num_pv_frames(n - start, hop))
    pvt = torch.empty((ch, Ft, bins, 2), dtype=torch.float32, device=dev)
    fa.analyze_dev(xt, ch, n - start, SR, w, hop, dft, pvt)
    torch.cuda.synchronize()
    skip = w // hop + 2
    assert F - (start // hop) == Ft
    assert bool(torch.equal(pvt[:, skip:].contiguous().view(torch.int32), pv[:, start // hop + skip:].contiguous().view(torch.int32)))


def test_repeated_runs_are_bit_identical():
    """the same buffers through the fused and the unfused round trip 40 times each: every PV and every output bit-identical to the
    first (no race between chains, no dependence on what an earlier step left in the workspace)"""
    import torch
    import flan_amd as fa
    dev = torch.device("cuda", 0)
    ch, n = 8, 20 * 48000
    F = int(fa.lib.flanhip_num_pv_frames(n, HOP))
    x = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(_p(x), ch, n, 99, None))
    ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, BINS, SR, AR, W), dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    first = None
    for fused in (True, False):
        for i in range(40):
            pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
            out = torch.empty((ch, F * HOP), dtype=torch.float32, device=dev)
            if fused:
                fa.analyze_dev_fused(x, ch, n, SR, W, HOP, DFT, pv, ws)
                fa.synthesize_dev_fused(pv, ch, F, BINS, SR, AR, W, out, ws, flag)
            else:
                fa.analyze_dev(x, ch, n, SR, W, HOP, DFT, pv)
                fa.synthesize_dev(pv, ch, F, BINS, SR, AR, W, out, ws, flag)
            torch.cuda.synchronize()
            if first is None:
                first = (pv.clone(), out.clone())
            else:
                assert bool(torch.equal(pv.view(torch.int32), first[0].view(torch.int32))), (fused, i)
                assert bool(torch.equal(out.view(torch.int32), first[1].view(torch.int32))), (fused, i)
    assert int(flag.item()) == 0


def test_config2_full_size_parity_against_the_oracle():
    """BASELINE config 2 in full (stereo 60 s 48 kHz white noise, convert_to_PV(2048,512,2048) -> convert_to_audio): the oracle runs
    the whole thing in a few seconds, so P1 / P2 / P3 are checked at the real size, not only on the head.
    P1 and P2 meet north_star's 1e-5.  P3 (the composite) does NOT on this input: 9.1e-5 RMS measured, asserted as a documented floor
    (<= 2e-4) -- ~0.9 % of the bins carry an f one fp32 step from the oracle's and 60 s of synthesis integrates it; the reference moves
    by 8.9e-5 itself when only its FFT backend changes.  Config 1 (5 s sine) does meet 1e-5 as a composite (8e-7)."""
    import flan_amd as fa
    x = O.noise(2, 60 * 48000, seed=1234)
    pv_ref = O.analyze(x, SR, W, HOP, DFT)
    pv_got = fa.analyze(x, SR, W, HOP, DFT)
    m_r, m_g = pv_ref[..., 0].astype(np.float64), pv_got[..., 0].astype(np.float64)
    rel_m = np.sqrt(np.sum((m_g - m_r) ** 2) / np.sum(m_r ** 2))
    same_f = np.mean(pv_ref[..., 1].view(np.uint32) == pv_got[..., 1].view(np.uint32))
    df = pv_got[..., 1].astype(np.float64) - pv_ref[..., 1]
    df -= np.rint(df / AR) * AR
    wrms_f = np.sqrt(np.sum(m_r ** 2 * df ** 2) / np.sum(m_r ** 2))
    ar = np.float32(SR) / np.float32(HOP)
    out_ref, _ = O.synthesize(pv_ref, SR, ar, W)
    out_same, flag = fa.synthesize(pv_ref, SR, ar, W)                       # P2: identical PV in
    out_got, _ = fa.synthesize(pv_got, SR, ar, W)                           # P3: composite
    p2 = np.sqrt(np.mean((out_same.astype(np.float64) - out_ref) ** 2))
    p3 = np.sqrt(np.mean((out_got.astype(np.float64) - out_ref) ** 2))
    print("\n[config 2, full size: %d frames] P1 rel_m=%.2e  wrms df=%.2e Hz  f bit-identical=%.4f   P2 rms=%.2e   P3 rms=%.2e"
          % (pv_ref.shape[0] * pv_ref.shape[1], rel_m, wrms_f, same_f, p2, p3))
    assert flag == 0
    assert rel_m <= 1e-5 and wrms_f <= 5e-4 and same_f >= 0.985          # measured 1.05e-7, 5e-5 Hz, 0.9913
    assert p2 <= 1e-5
    assert p3 <= 2e-4           # the documented exception to the 1e-5 (measured 9.1e-5; reference FFT-swap self-noise 8.9e-5)
    assert p3 > 1e-5            # should this composite ever meet the tolerance, drop the exception instead of keeping a loose bound


def test_config3_full_length_parity_against_the_oracle():
    """BASELINE config 3 at full length (60 s, stretch x2 by a sampled factor grid), on two of its eight channels (channels are
    independent): analysis -> stretch -> synthesis, every stage against the oracle fed with the oracle's previous stage"""
    import flan_amd as fa
    x = O.noise(2, 60 * 48000, seed=77)
    pv_ref = O.analyze(x, SR, W, HOP, DFT)
    F, bins = pv_ref.shape[1], pv_ref.shape[2]
    mod = O.stretch_map(np.full((F, bins), 2.0, np.float32), SR, HOP)
    st_ref = O.modify_time(pv_ref, SR, HOP, mod)
    st_got = fa.modify_time(pv_ref, SR, HOP, mod)
    assert st_got.shape == st_ref.shape == (2, 2 * F, bins, 2)
    assert np.array_equal(st_got.view(np.uint32), st_ref.view(np.uint32))       # P4: bit for bit
    ar = np.float32(SR) / np.float32(HOP)
    out_ref, _ = O.synthesize(st_ref, SR, ar, W)
    out_got, flag = fa.synthesize(st_ref, SR, ar, W)
    p2 = np.sqrt(np.mean((out_got.astype(np.float64) - out_ref) ** 2))
    print("\n[config 3, full length: %d -> %d frames per channel] stretch bit-identical; P2 rms=%.2e" % (F, 2 * F, p2))
    assert flag == 0 and p2 <= 1e-5


def test_config3_all_eight_channels_device_resident():
    """BASELINE config 3 WHOLE (8 ch x 60 s, stretch x2) through the device-resident entry points bench.py times (analysis -> stretch with
    the pre-pass hand-over -> synthesis), held to the two-channel run the oracle checks above by properties that need no CPU run of 8
    channels: channels are independent (AudioPV.cpp:41,44,108,111; PVModify.cpp:319), so (i) every pair of channels of the 8-channel job
    must equal the same pair run as a job of its own BIT FOR BIT -- PV, stretched PV and audio -- although chain lengths and block shapes
    differ between the two jobs (8 x 5626 frames cut into 2048 chains of 22, 2 x 5626 into 2048 chains of 6); (ii) channels 0 and 1 are the
    oracle-checked input of test_config3_full_length_parity_against_the_oracle; (iii) the stretched PV doubles the frame count and the
    audio its length; (iv) the fused hand-over and the plain calls agree."""
    import ctypes
    import torch
    import flan_amd as fa
    lib, vp = fa.lib, ctypes.c_void_p
    dev = torch.device("cuda", 0)
    fa.check(lib.flanhip_set_device(0))
    P = lambda t: vp(t.data_ptr())
    n, BINS = 60 * 48000, DFT // 2 + 1
    F = int(lib.flanhip_num_pv_frames(n, HOP))
    Fo = 2 * F
    ar = np.float32(SR) / np.float32(HOP)
    x8 = torch.from_numpy(O.noise(8, n, seed=77)).to(dev)
    assert np.array_equal(x8[:2].cpu().numpy(), O.noise(2, n, seed=77))               # (ii): the same first two channels

    def run(x, fused):
        ch = x.shape[0]
        pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
        grid = torch.empty((F, BINS), dtype=torch.float32, device=dev)
        dmax = torch.empty(1, dtype=torch.float32, device=dev)
        st = torch.empty((ch, Fo, BINS, 2), dtype=torch.float32, device=dev)
        out = torch.empty((ch, Fo * HOP), dtype=torch.float32, device=dev)
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, Fo, BINS, SR, ar, W), dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        fa.check(lib.flanhip_analyze_dev(P(x), ch, n, SR, W, HOP, DFT, P(pv), None))
        fa.check(lib.flanhip_fill_dev(P(grid), F * BINS, 2.0, None))
        fa.check(lib.flanhip_stretch_map_dev(P(grid), F, BINS, SR, HOP, P(dmax), None))
        if fused:
            fa.check(lib.flanhip_modify_time_dev_fused(P(pv), ch, F, BINS, SR, ar, P(grid), Fo, P(st), W, P(ws), None))
            fa.check(lib.flanhip_synthesize_dev_fused_checked(P(st), ch, Fo, BINS, SR, ar, W, P(out), P(ws), P(flag), None))
        else:
            fa.check(lib.flanhip_modify_time_dev(P(pv), ch, F, BINS, SR, HOP, P(grid), Fo, P(st), None))
            fa.check(lib.flanhip_synthesize_dev(P(st), ch, Fo, BINS, SR, ar, W, P(out), P(ws), P(flag), None))
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        return pv, st, out

    pv8, st8, out8 = run(x8, True)
    assert st8.shape == (8, 2 * F, BINS, 2) and out8.shape == (8, 2 * F * HOP)         # (iii)
    for c in range(0, 8, 2):                                                           # (i)
        pv2, st2, out2 = run(x8[c:c + 2].contiguous(), True)
        assert torch.equal(pv8[c:c + 2].view(torch.int32), pv2.view(torch.int32)), c
        assert torch.equal(st8[c:c + 2].view(torch.int32), st2.view(torch.int32)), c
        d = (out8[c:c + 2].double() - out2.double()).abs().max().item()
        assert d <= 2e-6, (c, d)                                                        # chain boundaries differ: fp32 re-association there only
        del pv2, st2, out2
    pvp, stp, outp = run(x8, False)                                                    # (iv)
    assert torch.equal(st8.view(torch.int32), stp.view(torch.int32))
    assert (out8.double() - outp.double()).abs().max().item() <= 2e-6


def test_config5_ten_seconds_parity_against_the_oracle():
    """BASELINE config 5 (stereo 96 kHz -> resample to 48 kHz -> convert_to_PV -> shape f + 100 Hz -> convert_to_audio), 10 of
    its 60 seconds (the scalar oracle FIR needs ~0.4 s per second of audio): stage by stage against the oracle"""
    import flan_amd as fa
    x96 = O.noise(2, 10 * 96000, seed=1234)
    x48_r = O.resample_2to1(x96, 96000.0, SR)
    x48_g = fa.resample(x96, 96000.0, SR)
    same = np.mean(x48_g.view(np.uint32) == x48_r.view(np.uint32))
    assert x48_g.shape == x48_r.shape == (2, 10 * 48000) and same >= 0.9999
    pv_r = O.analyze(x48_r, SR, W, HOP, DFT)
    pv_g = fa.analyze(x48_r, SR, W, HOP, DFT)
    m_r, m_g = pv_r[..., 0].astype(np.float64), pv_g[..., 0].astype(np.float64)
    rel_m = np.sqrt(np.sum((m_g - m_r) ** 2) / np.sum(m_r ** 2))
    sh_r = O.shape_affine(pv_r, SR, 1.0, 0.0, 1.0, 100.0, False)
    sh_g = fa.shape_affine(pv_r, SR, 1.0, 0.0, 1.0, 100.0, False)
    assert np.array_equal(sh_g.view(np.uint32), sh_r.view(np.uint32))
    ar = np.float32(SR) / np.float32(HOP)
    out_r, _ = O.synthesize(sh_r, SR, ar, W)
    out_g, flag = fa.synthesize(sh_r, SR, ar, W)
    p2 = np.sqrt(np.mean((out_g.astype(np.float64) - out_r) ** 2))
    print("\n[config 5, 10 s] resample bit-identical=%.5f  P1 rel_m=%.2e  shape bit-identical  P2 rms=%.2e" % (same, rel_m, p2))
    assert rel_m <= 1e-5 and p2 <= 1e-5 and flag == 0


def test_arranging_methods_at_the_config3_size():
    """8 ch x 60 s through the frame-selecting / warping methods: identities that need no CPU run, and repeatability of the kernels
    that resolve conflicts with atomics (modify, add_harmonics)"""
    import torch
    import flan_amd as fa
    lib = fa.lib
    dev = torch.device("cuda", 0)
    ch, n = 8, 60 * 48000
    F = int(lib.flanhip_num_pv_frames(n, HOP))
    x = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(_p(x), ch, n, 99, None))
    pv = torch.empty((ch, F, BINS, 2), dtype=torch.float32, device=dev)
    fa.analyze_dev(x, ch, n, SR, W, HOP, DFT, pv)
    torch.cuda.synchronize()

    def same(a, b):
        return bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))

    # cut into three pieces and joined again = the input minus its last frame (cut_frames clamps its end to F-1, PV.cpp:653)
    cuts = [(0, 1000), (1000, 4000), (4000, F)]
    joined = torch.zeros((ch, F - 1, BINS, 2), dtype=torch.float32, device=dev)
    at = 0
    for a, b in cuts:
        s0, c0 = C.c_int32(), C.c_int32()
        fa.check(lib.flanhip_cut_frames_range(F, a, b, C.byref(s0), C.byref(c0)))
        piece = torch.empty((ch, c0.value, BINS, 2), dtype=torch.float32, device=dev)
        fa.check(lib.flanhip_cut_frames_dev(_p(pv), ch, F, BINS, s0.value, c0.value, _p(piece), None))
        fa.check(lib.flanhip_place_frames_dev(_p(piece), ch, c0.value, BINS, _p(joined), ch, F - 1, BINS, at, None))
        at += c0.value
    torch.cuda.synchronize()
    assert at == F - 1 and same(joined, pv[:, :F - 1].contiguous())

    # freeze with no pauses = a copy; with pauses every output frame is the input frame the plan names
    src = fa.freeze_plan(F, SR, HOP, [], [])
    assert np.array_equal(src, np.arange(F))
    src = fa.freeze_plan(F, SR, HOP, [7.0, 31.5, 59.9], [0.5, 2.0, 1.0])
    d_src = torch.from_numpy(src).to(dev)
    fz = torch.empty((ch, len(src), BINS, 2), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_select_frames_dev(_p(pv), ch, F, BINS, _p(d_src), len(src), _p(fz), None))
    torch.cuda.synchronize()
    valid = d_src >= 0
    assert same(fz[:, valid], pv[:, d_src[valid].long()]) and not bool(fz[:, ~valid].any())

    # get_frame on a whole frame = that frame
    one = torch.empty((ch, 1, BINS, 2), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_get_frame_dev(_p(pv), ch, F, BINS, 4321.0, 0, _p(one), None))
    torch.cuda.synchronize()
    assert same(one[:, 0], pv[:, 4321])

    # modify with the identity map: twice, bit for bit; and wherever frame -> time -> frame is exact in fp32 (93.75 frames per second:
    # not everywhere) a grid point is a corner of its quad and comes back as it went in
    t = (torch.arange(F, device=dev, dtype=torch.float32) / (SR / HOP))[:, None].expand(F, BINS)
    f = (torch.arange(BINS, device=dev, dtype=torch.float32) * (SR / DFT))[None, :].expand(F, BINS)
    ident = torch.stack([t, f], dim=-1).contiguous()
    in_f = pv[..., 1].contiguous()
    outs = []
    for rep in range(2):
        o = torch.empty((ch, F - 1, BINS, 2), dtype=torch.float32, device=dev)
        fa.check(lib.flanhip_modify_dev(_p(pv), ch, F, BINS, SR, HOP, _p(ident), _p(in_f), 0, F - 1, _p(o), None))
        outs.append(o)
    torch.cuda.synchronize()
    assert same(outs[0], outs[1])
    tf = (t[:, 0] * SR) / HOP
    exact = tf == torch.arange(F, device=dev, dtype=torch.float32)
    rows = (exact[:-1] & exact[1:]).nonzero().flatten()
    rows = rows[rows < F - 1]
    assert len(rows) > F // 4
    assert same(outs[0][:, rows][:, :, :BINS - 1].contiguous(), pv[:, rows][:, :, :BINS - 1].contiguous()) and not bool(outs[0][:, :, BINS - 1].any())

    # add_harmonics twice: the same bits (placement through atomics must not depend on timing)
    series = torch.full((F, BINS), 0.5, dtype=torch.float32, device=dev)
    hs = []
    for rep in range(2):
        o = torch.empty_like(pv)
        fa.check(lib.flanhip_harmonic_scale_dev(_p(pv), ch, F, BINS, SR, _p(series), BINS, 1, _p(o), None))
        hs.append(o)
    torch.cuda.synchronize()
    assert same(hs[0], hs[1]) and bool(hs[0].any())

    # stretch_spline with steps of one reproduces the knots (each evaluated from the segment before it: rounding only)
    steps = np.ones(F - 1, np.uint32)
    sp = torch.empty((ch, F - 1, BINS, 2), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_stretch_spline_dev(_p(pv), ch, F, BINS, C.c_void_p(steps.ctypes.data), F - 1, _p(sp), None))
    torch.cuda.synchronize()
    assert same(sp[:, 0], pv[:, 0])
    err = (sp - pv[:, :F - 1]).abs().max().item()
    assert err < 2e-2 * pv.abs().max().item() * 1e-3 + 1e-2, err
