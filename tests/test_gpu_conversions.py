"""GPU parity of Audio::convert_to_PV / PV::convert_to_audio (HIP, through the C ABI) against the CPU oracle.

Tolerances (BASELINE north_star: <= 1e-5 RMS vs the reference, stage-wise -- SURVEY 8d):
  P1 analysis : ||dm|| / ||m|| <= 1e-5 ;  magnitude-weighted RMS of df <= 2e-3 Hz and most f bit-identical
                (the reference's own f flips by one fp32 step in ~0.8 % of bins when only its FFT backend changes)
  P2 synthesis: identical PV in -> RMS(audio diff) <= 1e-5 of unit scale (typically ~1e-7)
  P3 composite: round trip on the 5 s sine (config 1) <= 1e-5 RMS
  NOT met, and said so: the composite on LONG NOISE (config 2).  north_star's 1e-5 holds stage-wise there (P1, P2) but the round trip
  of 60 s of noise differs from the oracle's by 9.1e-5 RMS (10 s: ~3e-5): ~0.9 % of the bins get an f one fp32 step away from the
  oracle's under ANY change of FFT arithmetic, and synthesis integrates f for the whole signal.  The reference itself moves by 8.9e-5
  when only its FFT backend is swapped.  Those tests assert the measured floor with head-room (<= 2e-4) and print the value;
  they do not claim the 1e-5.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import flan_amd
    assert flan_amd.lib.flanhip_device_count() > 0
    return flan_amd


def p1_metrics(pv_gpu, pv_ref, analysis_rate):
    m_g, f_g = pv_gpu[..., 0].astype(np.float64), pv_gpu[..., 1].astype(np.float64)
    m_r, f_r = pv_ref[..., 0].astype(np.float64), pv_ref[..., 1].astype(np.float64)
    rel_m = np.sqrt(np.sum((m_g - m_r) ** 2) / max(np.sum(m_r ** 2), 1e-300))
    df = f_g - f_r
    # a wrap decision taken the other way (delta/2pi within rounding of a half integer) moves f by exactly one
    # analysis_rate and is immaterial to synthesis (phase advance differs by one whole turn): fold it out, count it
    turns = np.rint(df / analysis_rate)
    df_folded = df - turns * analysis_rate
    w = m_r ** 2
    wrms_f = np.sqrt(np.sum(w * df_folded ** 2) / max(np.sum(w), 1e-300))
    # bins whose magnitude is rounding noise (a pure tone leaves most bins at ~1e-5 of the peak) have noise for a
    # phase in the reference too: bit-level statistics are taken over the significant bins only
    sig = m_r > 1e-4 * max(m_r.max(), 1e-300)
    eq = pv_gpu[..., 1].view(np.uint32) == pv_ref[..., 1].view(np.uint32)
    same = float(np.mean(eq[sig])) if sig.any() else 1.0
    return rel_m, wrms_f, same, int(np.count_nonzero(turns[sig]))


CASES = [
    # name, channels, n, window, hop, dft, kind
    ("sine_5s_cfg1", 1, 240000, 2048, 512, 2048, "sine"),
    ("noise_stereo", 2, 48000, 2048, 512, 2048, "noise"),
    ("noise_dft4096_hop128", 1, 20000, 2048, 128, 4096, "noise"),
    # dft 4096, window <= 2048: two 1024-point transforms per frame (pv_kernels_eo.h); hop 256 / 512 / 1024 also in synthesis
    ("noise_dft4096_hop512", 2, 50000, 2048, 512, 4096, "noise"),
    ("noise_dft4096_hop256", 1, 30000, 2048, 256, 4096, "noise"),
    ("noise_dft4096_hop1024_3ch", 3, 41000, 2048, 1024, 4096, "noise"),
    ("dft4096_win1024_hop256", 1, 20000, 1024, 256, 4096, "noise"),
    ("dft4096_short", 2, 3000, 2048, 512, 4096, "noise"),
    ("dft4096_one_frame", 1, 300, 2048, 512, 4096, "noise"),
    ("dft4096_win3000", 1, 30000, 3000, 512, 4096, "noise"),
    # dft 4096 with windows above 2048 (window = dft is the plain STFT call): the WBIG team kernels
    ("dft4096_win4096_hop1024", 2, 90000, 4096, 1024, 4096, "noise"),
    ("dft4096_win4096_hop512", 1, 60000, 4096, 512, 4096, "noise"),
    ("dft4096_win4096_hop128", 1, 30000, 4096, 128, 4096, "noise"),
    ("dft4096_win3072_hop256", 1, 40000, 3072, 256, 4096, "noise"),
    ("dft4096_win4096_hop256", 2, 50000, 4096, 256, 4096, "noise"),
    ("sine_dft4096_win4096", 1, 48000, 4096, 1024, 4096, "sine"),
    ("dft4096_win4096_short", 2, 3000, 4096, 512, 4096, "noise"),
    ("dft4096_win4096_one_frame", 1, 300, 4096, 1024, 4096, "noise"),
    ("dft4096_win2304_hop128", 1, 20000, 2304, 128, 4096, "noise"),
    # dft 4096 off the tuned grid (round 5): the team synthesis with its overlap-add accumulator as a ring in LDS (odd and even hops, windows that are no
    # multiple of 256, windows above 2048: three teams per block)
    ("dft4096_hop441", 2, 60000, 2048, 441, 4096, "noise"),
    ("dft4096_win2000_hop500", 1, 50000, 2000, 500, 4096, "noise"),
    ("dft4096_win4000_hop1000", 2, 90000, 4000, 1000, 4096, "noise"),
    ("dft4096_win1999_hop333", 1, 30000, 1999, 333, 4096, "noise"),
    ("dft4096_win3001_hop750", 1, 40000, 3001, 750, 4096, "noise"),
    ("dft4096_hop_eq_window_1000", 1, 30000, 1000, 1000, 4096, "noise"),
    ("sine_dft4096_hop512", 1, 48000, 2048, 512, 4096, "sine"),
    ("ragged_len", 3, 12345, 2048, 512, 2048, "noise"),
    ("one_frame", 1, 100, 2048, 512, 2048, "noise"),
    ("short_two_frames", 2, 600, 2048, 512, 2048, "noise"),
    ("zeros", 1, 5000, 2048, 512, 2048, "zeros"),
    ("small_dft256", 2, 9000, 256, 64, 256, "noise"),
    ("dft512_win400", 1, 9000, 400, 100, 512, "noise"),
    ("dft1024", 1, 30000, 1024, 256, 1024, "noise"),
    # dft 1024 / 512: the one-wavefront-per-chain register kernels (pv_kernels_v3.h); synthesis hops 128 / 256 / 512 / 1024, windows that are multiples of 128
    ("dft1024_hop128_stereo", 2, 41000, 1024, 128, 1024, "noise"),
    ("dft1024_hop512_win768", 1, 50000, 768, 512, 1024, "noise"),
    ("dft1024_ragged_3ch", 3, 12345, 1024, 256, 1024, "noise"),
    ("dft1024_one_frame", 1, 100, 1024, 256, 1024, "noise"),
    ("dft1024_sine", 1, 48000, 1024, 256, 1024, "sine"),
    ("dft1024_zeros", 1, 5000, 1024, 256, 1024, "zeros"),
    ("dft512", 2, 30000, 512, 128, 512, "noise"),
    ("dft512_hop256_win384", 1, 30000, 384, 256, 512, "noise"),
    ("dft512_short", 2, 700, 512, 128, 512, "noise"),
    ("dft512_sine", 1, 48000, 512, 128, 512, "sine"),
    # ... and off their grid (hops that are no multiple of 128, windows that are none): the tuned analysis, the synthesis with its accumulator as an LDS ring
    ("dft1024_win1000_hop250", 2, 30000, 1000, 250, 1024, "noise"),
    ("dft1024_hop300", 1, 30000, 1024, 300, 1024, "noise"),
    ("dft1024_win999_hop333_ragged", 3, 12345, 999, 333, 1024, "noise"),
    ("dft512_win500_hop125", 2, 20000, 500, 125, 512, "noise"),
    ("dft512_hop100", 1, 20000, 512, 100, 512, "noise"),
    # dft 512 / 256 on the grid of the kernels with several chains per wavefront (pv_kernels_sub.h, round 6: 32 / 16 lanes per chain): hops of 1, 2, 4, 8 steps of
    # 64 / 32 samples, windows that are multiples of a step; short last chains, spare lane groups, one frame
    ("dft256_ragged_3ch", 3, 12345, 256, 64, 256, "noise"),
    ("dft256_hop32", 1, 20000, 256, 32, 256, "noise"),
    ("dft256_hop128_win192", 2, 20000, 192, 128, 256, "noise"),
    ("dft256_hop256", 1, 20000, 256, 256, 256, "noise"),
    ("dft256_one_frame", 1, 50, 256, 64, 256, "noise"),
    ("dft256_sine", 1, 48000, 256, 64, 256, "sine"),
    ("dft256_zeros", 1, 5000, 256, 64, 256, "zeros"),
    ("dft256_5ch_long", 5, 200000, 256, 64, 256, "noise"),
    # dft 128 on the same kernels: eight lanes per chain, eight chains per wavefront (the last pass only carries the elements back to their lanes)
    ("dft128_ragged_3ch", 3, 12345, 128, 32, 128, "noise"),
    ("dft128_hop16", 1, 20000, 128, 16, 128, "noise"),
    ("dft128_hop64_win96", 2, 20000, 96, 64, 128, "noise"),
    ("dft128_hop128", 1, 20000, 128, 128, 128, "noise"),
    ("dft128_one_frame", 1, 50, 128, 32, 128, "noise"),
    ("dft128_sine", 1, 48000, 128, 32, 128, "sine"),
    ("dft128_zeros", 1, 5000, 128, 32, 128, "zeros"),
    ("dft128_5ch_long", 5, 300000, 128, 32, 128, "noise"),
    ("dft512_hop64", 1, 30000, 512, 64, 512, "noise"),
    ("dft512_hop512_win512", 2, 40000, 512, 512, 512, "noise"),
    ("dft512_7ch_ragged", 7, 54321, 512, 128, 512, "noise"),
    ("dft512_one_frame", 1, 100, 512, 128, 512, "noise"),
    ("dft512_zeros", 1, 5000, 512, 128, 512, "zeros"),
    ("dft8192", 1, 40000, 4096, 1024, 8192, "noise"),
    # dft 8192 / 16384 on the grid of the team kernels (pv_kernels_team.h, round 6: teams of 4 / 8 wavefronts, a 1024-point register transform each):
    # window = 4 / 8 / 16 steps of 512 (1024) samples, hops of 1 / 2 / 4 / 8 steps
    ("dft8192_win8192_hop2048_stereo_ragged", 2, 123457, 8192, 2048, 8192, "noise"),
    ("dft8192_win2048_hop512", 1, 40000, 2048, 512, 8192, "noise"),
    ("dft8192_win2048_hop1024_3ch", 3, 30000, 2048, 1024, 8192, "noise"),
    ("dft8192_win4096_hop512", 1, 40000, 4096, 512, 8192, "noise"),
    ("dft8192_win4096_hop2048", 2, 50000, 4096, 2048, 8192, "noise"),
    ("dft8192_win8192_hop1024", 1, 60000, 8192, 1024, 8192, "noise"),
    ("dft8192_win8192_hop4096", 1, 90000, 8192, 4096, 8192, "noise"),
    ("dft8192_sine", 1, 96000, 8192, 2048, 8192, "sine"),
    ("dft8192_one_frame", 1, 300, 8192, 2048, 8192, "noise"),
    ("dft8192_short", 2, 5000, 4096, 1024, 8192, "noise"),
    ("dft8192_zeros", 1, 20000, 8192, 2048, 8192, "zeros"),
    ("dft8192_off_grid_win4000_hop1000", 1, 40000, 4000, 1000, 8192, "noise"),         # (the round-1 kernels)
    ("dft16384_win16384_hop4096_stereo_ragged", 2, 234567, 16384, 4096, 16384, "noise"),
    ("dft16384_win8192_hop2048", 1, 90000, 8192, 2048, 16384, "noise"),
    ("dft16384_win4096_hop2048", 2, 50000, 4096, 2048, 16384, "noise"),
    ("dft16384_win8192_hop1024", 1, 60000, 8192, 1024, 16384, "noise"),
    ("dft16384_win8192_hop4096", 1, 90000, 8192, 4096, 16384, "noise"),
    ("dft16384_win16384_hop2048", 1, 120000, 16384, 2048, 16384, "noise"),
    ("dft16384_win16384_hop8192", 1, 150000, 16384, 8192, 16384, "noise"),
    ("dft16384_sine", 1, 96000, 4096, 1024, 16384, "sine"),
    ("dft16384_one_frame", 1, 300, 4096, 1024, 16384, "noise"),
    ("dft16384_zeros", 1, 20000, 4096, 1024, 16384, "zeros"),
    ("dft16384_off_grid_win4096_hop512", 1, 30000, 4096, 512, 16384, "noise"),          # (the mixed-radix kernels)
    ("dft64", 1, 3000, 64, 16, 64, "noise"),
    ("dft32_win32", 1, 1000, 32, 8, 32, "noise"),
    ("hop_eq_window", 1, 20000, 1024, 1024, 1024, "noise"),
    ("hop_gt_window", 1, 20000, 512, 700, 1024, "noise"),
    ("odd_hop", 1, 20000, 2048, 333, 2048, "noise"),
    ("dft2048_win2000_hop500_stereo", 2, 30000, 2000, 500, 2048, "noise"),
    ("dft2048_win1801_hop450_ragged", 3, 12345, 1801, 450, 2048, "noise"),
    # dft sizes without power-of-two kernels (any even size goes to FFTW in the reference, FFTHelper.cpp:16-26): the mixed-radix kernels
    # (pv_kernels_mr.h: half the size a product of 2, 3, 5, 7, 11, 13, at most 8192) and, for the rest (2998 = 2 x 1499), the direct sums
    # (pv_kernels_any.h)
    ("dft2002_radices_7_11_13", 1, 20000, 1024, 256, 2002, "noise"),
    ("dft1920_win1024", 2, 20000, 1024, 256, 1920, "noise"),
    ("dft12000_in_place_odd_radices", 1, 40000, 2048, 512, 12000, "noise"),
    ("dft6000_hop300", 1, 30000, 2400, 300, 6000, "noise"),
    ("dft3000", 1, 30000, 2048, 512, 3000, "noise"),
    ("dft3000_stereo_ragged", 2, 12345, 2048, 512, 3000, "noise"),
    ("dft16384_win4096", 1, 60000, 4096, 1024, 16384, "noise"),
    ("dft1000_win600", 2, 9000, 600, 150, 1000, "noise"),
    # ... and, since round 5, Bluestein's chirp-z form for sizes with a larger prime factor (pv_kernels_bs.h: 64 <= dft / 2 <= 4096; two layouts)
    ("dft2998_prime_factor", 1, 9000, 1024, 256, 2998, "noise"),
    ("dft2998_full_window_stereo_ragged", 2, 12345, 2998, 750, 2998, "noise"),
    ("dft5998_chirp_in_place", 1, 30000, 2048, 512, 5998, "noise"),
    ("dft2018_chirp", 1, 20000, 1024, 256, 2018, "noise"),
    ("dft134_chirp_small", 2, 3000, 128, 32, 134, "noise"),
    ("dft4094_sine", 1, 48000, 2048, 512, 4094, "sine"),
    ("dft2998_one_frame", 1, 100, 1024, 256, 2998, "noise"),
    ("dft2998_zeros", 1, 5000, 1024, 256, 2998, "zeros"),
    # ... and, since round 5, sizes above 16384 with a power-of-two factor of at least 1024 in their half (pv_kernels_big.h: residue pairs of C1 x C2)
    ("dft32768_win4096", 1, 60000, 4096, 1024, 32768, "noise"),
    ("dft65536_win2048_stereo_ragged", 2, 23456, 2048, 512, 65536, "noise"),
    ("dft24576_three_residues", 1, 40000, 4096, 1024, 24576, "noise"),
    ("dft32768_win12000_two_segments", 1, 60000, 12000, 3000, 32768, "noise"),
    ("dft18432_c2_1024", 1, 30000, 2048, 512, 18432, "noise"),
    ("dft32768_sine", 1, 48000, 4096, 1024, 32768, "sine"),
    ("dft32768_one_frame", 1, 100, 4096, 1024, 32768, "noise"),
    # ... with windows whose overlap-add ring does not fit the LDS beside the transforms (above ~14 k samples: the ring in the workspace, round 6) -- window = dft
    # = 32768 at hop 8192 is the PaulStretch-style call
    ("dft32768_win32768_hop8192", 1, 300000, 32768, 8192, 32768, "noise"),
    ("dft65536_win20000_hop5000_stereo", 2, 150000, 20000, 5000, 65536, "noise"),
    ("dft32768_win16384_hop1024", 1, 60000, 16384, 1024, 32768, "noise"),
    # ... and (round 6) the sizes above 16384 whose half holds less than 2^10 as a power of two: C2 = any product of 2 ... 13 up to 4096 (bs_plan.h: `mixed`) --
    # 20000 = 2 x 4 x 2500, 44100 = 2 x 6 x 3675, 48000 = 2 x 6 x 4000 (two segments), 22050 (C1, C2 odd: the bin C / 2 mirrors itself), 100000 = 2 x 16 x 5^5,
    # 17836 = 2 x 7 x ( 2 x 7^2 x 13 ), 19602 = 2 x 3 x ( 3^3 x 11^2 )
    ("dft20000_win4096", 1, 60000, 4096, 1024, 20000, "noise"),
    ("dft44100_stereo_ragged", 2, 23456, 2048, 512, 44100, "noise"),
    ("dft48000_win12000_two_segments", 1, 60000, 12000, 3000, 48000, "noise"),
    ("dft22050_odd_halves", 1, 40000, 4096, 1024, 22050, "noise"),
    ("dft100000_win4096", 1, 40000, 4096, 1024, 100000, "noise"),
    ("dft17836_radix13", 1, 30000, 2048, 512, 17836, "noise"),
    ("dft19602_radix11", 1, 30000, 2048, 512, 19602, "noise"),
    ("dft20000_win20000_hop5000", 1, 200000, 20000, 5000, 20000, "noise"),
    ("dft20000_sine", 1, 48000, 4096, 1024, 20000, "sine"),
    ("dft20000_one_frame", 1, 100, 4096, 1024, 20000, "noise"),
    # the team kernels at hop = window / 16 (the reference API's default ratio scaled up): half a step of 64 R samples at window = dft / 2 and dft / 4, a whole one at window = dft
    ("dft8192_win4096_hop256_default_ratio", 1, 90000, 4096, 256, 8192, "noise"),
    ("dft16384_win8192_hop512_default_ratio", 1, 150000, 8192, 512, 16384, "noise"),
    ("dft8192_win2048_hop256_half_step_stereo_ragged", 2, 40123, 2048, 256, 8192, "noise"),
    ("dft16384_win4096_hop512_half_step", 1, 70000, 4096, 512, 16384, "noise"),
    ("dft8192_win8192_hop512", 1, 120000, 8192, 512, 8192, "noise"),
    ("dft16384_win16384_hop1024", 1, 200000, 16384, 1024, 16384, "noise"),
    ("dft8192_half_step_one_frame", 1, 100, 4096, 256, 8192, "noise"),
    ("dft8192_half_step_sine", 1, 48000, 4096, 256, 8192, "sine"),
    # ... and with the API's default hop 128 kept while the sizes grow: a quarter / an eighth of a step (16 / 8 lanes per hop, the accumulator moving on through ds_bpermute),
    # windows of two steps
    ("dft8192_win2048_hop128_quarter_step", 1, 50000, 2048, 128, 8192, "noise"),
    ("dft16384_win2048_hop128_eighth_step", 1, 50000, 2048, 128, 16384, "noise"),
    ("dft8192_win4096_hop128_stereo_ragged", 2, 30123, 4096, 128, 8192, "noise"),
    ("dft16384_win4096_hop256_quarter_step", 1, 60000, 4096, 256, 16384, "noise"),
    ("dft8192_win1024_hop128_two_steps", 1, 30000, 1024, 128, 8192, "noise"),
    ("dft16384_win2048_hop1024_two_steps", 1, 60000, 2048, 1024, 16384, "noise"),
    ("dft16384_win2048_hop512_two_steps_half", 1, 60000, 2048, 512, 16384, "noise"),
    ("dft8192_win8192_hop128_sixteen_steps_quarter", 1, 60000, 8192, 128, 8192, "noise"),
    ("dft8192_quarter_step_one_frame", 1, 100, 2048, 128, 8192, "noise"),
    ("dft16384_win16384_hop512_sixteen_steps_half", 1, 150000, 16384, 512, 16384, "noise"),
    ("dft8192_win8192_hop256_sixteen_steps_half_stereo", 2, 60123, 8192, 256, 8192, "noise"),
    # the mixed-radix kernels with their overlap-add ring in the workspace (windows the ring does not fit the LDS with: above ~6000 samples at dft 16384 off the team grid,
    # above ~10000 at dft 15000 -- direct sums until round 6: 0.8 s for 8 ch x 60 s at ( 8192, 512, 16384 ))
    ("dft16384_win8192_hop256_ring_in_workspace", 1, 120000, 8192, 256, 16384, "noise"),
    ("dft16384_win10000_hop2500_stereo_ragged", 2, 90123, 10000, 2500, 16384, "noise"),
    ("dft16384_win16384_hop4100", 1, 200000, 16384, 4100, 16384, "noise"),
    ("dft15000_win14000_hop3500", 1, 150000, 14000, 3500, 15000, "noise"),
    ("dft16384_win8192_hop256_one_frame", 1, 100, 8192, 256, 16384, "noise"),
    # Bluestein's form with its buffers in device memory (BsPlan::glob, round 6): sizes above 8192 with a prime factor above 13 in every divisor of their half --
    # 9998 = 2 x 4999, 10002 = 2 x 3 x 1667 (M = 16384), 30002 = 2 x 7 x 2143 (above 16384, no residue-pair plan: M = 32768) -- the direct sums until now
    ("dft9998_chirp_in_memory", 1, 60000, 4096, 1024, 9998, "noise"),
    ("dft9998_full_window_stereo_ragged", 2, 50123, 9998, 2499, 9998, "noise"),
    ("dft10002_win2000", 1, 30000, 2000, 500, 10002, "noise"),
    ("dft30002_win8000", 1, 60000, 8000, 2000, 30002, "noise"),
    ("dft9998_sine", 1, 48000, 4096, 1024, 9998, "sine"),
    ("dft9998_one_frame", 1, 100, 4096, 1024, 9998, "noise"),
    ("dft9998_zeros", 1, 20000, 4096, 1024, 9998, "zeros"),
    ("dft66", 1, 3000, 64, 16, 66, "noise"),
    ("dft6_win4", 1, 300, 4, 2, 6, "noise"),
    ("dft3000_sine", 1, 48000, 2048, 512, 3000, "sine"),
    ("dft3000_one_frame", 1, 100, 2048, 512, 3000, "noise"),
    ("dft3000_zeros", 1, 5000, 2048, 512, 3000, "zeros"),
]


def chirp_z_size(dft):
    """bs_plan.h: half the size with a prime factor above 13, 64 <= dft / 2 <= 4096 -- or (round 6, the layout in device memory) up to 131072 where no
    residue-pair plan exists (bs_plan_in_use)"""
    c = dft // 2
    if dft % 2 or c < 64 or c > 131072:
        return False
    if c > 4096 and dft > 16384 and _big_mixed_plan(dft) is not None:
        return False
    for r in (2, 3, 5, 7, 11, 13):
        while c % r == 0:
            c //= r
    return c != 1


def make_input(kind, ch, n):
    if kind == "sine":
        return O.sine(n)
    if kind == "zeros":
        return np.zeros((ch, n), np.float32)
    return O.noise(ch, n, seed=1234)


@pytest.mark.parametrize("name,ch,n,W,hop,dft,kind", CASES, ids=[c[0] for c in CASES])
def test_analysis_parity(fa, name, ch, n, W, hop, dft, kind):
    x = make_input(kind, ch, n)
    sr = 48000.0
    ref = O.analyze(x, sr, W, hop, dft)
    got = fa.analyze(x, sr, W, hop, dft)
    assert got.shape == ref.shape
    assert np.all(np.isfinite(got))
    rel_m, wrms_f, same, turns = p1_metrics(got, ref, sr / hop)
    print("\n[P1 %s] rel_m=%.3e wrms_df=%.3e Hz  bit-identical f=%.4f  whole-turn flips=%d" % (name, rel_m, wrms_f, same, turns))
    if kind == "zeros":
        assert np.array_equal(got[..., 0], ref[..., 0])
        return
    assert rel_m <= 1e-5
    # measured 2e-5 .. 3.3e-4 Hz (the largest at dft 32 / hop 8); one last-bit change of a phase is 1.2e-7 rad x analysis_rate / 2 pi Hz, so the
    # bound follows the analysis rate where that is extreme (hop 2: 24 kHz)
    # (a pure tone through a short transform: the bins at 1e-6 of the peak hold rounding noise for a phase in the oracle too, f there is anybody's within
    # an analysis rate, and with few bins per frame their m^2 df^2 is what the weighted figure consists of -- 7.2e-4 Hz at dft 512 from the tuned and 7.7e-4
    # from the generic kernels alike, tools/dbg_sine.py)
    # (the exception is dft 512's and dft 256's: 6.7e-4 .. 7.7e-4 and 6.3e-4 .. 6.5e-4 measured, from every kernel generation alike -- tools/dbg_sine_small.py)
    assert wrms_f <= (1e-3 if kind == "sine" and dft in (256, 512) else max(5e-4, 1e-7 * sr / hop))
    if kind == "noise":
        assert turns <= max(3, got[..., 0].size // 100000)
        # share of f words that are bit for bit the oracle's.  What is left differs by one rounding of the transform (two FFTs in two operation
        # orders); how many f words that flips grows with analysis_rate / bin width, hence the floors by shape (measured: 0.990-0.997 at
        # hop >= 512 and dft >= 2048, 0.984 at hop 256, 0.971 at hop 128 or dft 512, 0.96 / 0.90 / 0.86 at dft 256 / 64 / 32)
        floor = 0.985 if ( hop >= 512 and dft >= 2048 ) else 0.98 if ( hop >= 256 and dft >= 1024 ) else 0.965 if dft >= 512 else 0.95 if dft >= 256 else 0.85
        # Bluestein's form (pv_kernels_bs.h) runs TWO fp32 transforms of the next power of two above dft - 1 per frame, not one of dft / 2: with
        # the chirp products and the split in double it measures 0.9795-0.981 at hop 256 where one transform gives 0.984 (all in fp32: 0.978)
        if chirp_z_size(dft):
            floor -= 0.006
        # (hop = dft / 8 at the small sizes: twice the analysis rate per bin width of the hop = dft / 4 shapes the floors were taken on -- 0.927 at
        # (256, 32, 256), from the kernels of pv_kernels_sub.h and from their predecessors alike: test_sub_kernels_agree_with_their_predecessors)
        if dft <= 512 and hop * 8 <= dft:
            floor -= 0.03
        assert same >= floor


@pytest.mark.parametrize("name,ch,n,W,hop,dft,kind", CASES, ids=[c[0] for c in CASES])
def test_synthesis_parity(fa, name, ch, n, W, hop, dft, kind):
    """identical PV (the oracle's) into both synthesisers"""
    x = make_input(kind, ch, n)
    sr = 48000.0
    pv = O.analyze(x, sr, W, hop, dft)
    ar = np.float32(sr) / np.float32(hop)
    ref, flag_r = O.synthesize(pv, sr, ar, W)
    got, flag_g = fa.synthesize(pv, sr, ar, W)
    assert got.shape == ref.shape
    assert flag_g == flag_r == 0
    rms = float(np.sqrt(np.mean((got.astype(np.float64) - ref.astype(np.float64)) ** 2)))
    peak = float(np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64))))
    scale = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    print("\n[P2 %s] rms diff=%.3e  peak diff=%.3e  (signal rms %.3e)" % (name, rms, peak, scale))
    assert rms <= 1e-5


def test_roundtrip_config1(fa):
    """BASELINE config 1: mono 5 s 48 kHz sine -> convert_to_PV(2048,512,2048) -> convert_to_audio; composite P3."""
    x = O.sine(240000)
    sr = 48000.0
    pv_r = O.analyze(x, sr, 2048, 512, 2048)
    out_r, _ = O.synthesize(pv_r, sr, sr / 512, 2048)
    pv_g = fa.analyze(x, sr, 2048, 512, 2048)
    out_g, _ = fa.synthesize(pv_g, sr, sr / 512, 2048)
    rms = float(np.sqrt(np.mean((out_g.astype(np.float64) - out_r.astype(np.float64)) ** 2)))
    print("\n[P3 config1] composite rms diff=%.3e" % rms)
    assert rms <= 1e-5
    # SURVEY 8c anchors hold on the GPU path too
    assert out_g[0, 1000] == pytest.approx(0.43333316, rel=1e-5)
    assert np.sum(out_g.astype(np.float64) ** 2) == pytest.approx(30006.053, rel=1e-5)


def test_roundtrip_noise_composite(fa):
    """The composite on noise does NOT meet north_star's 1e-5 (see the module docstring): 10 s here, measured ~3e-5, asserted against
    the documented floor with head-room.  (60 s: 9.1e-5, tests/test_gpu_full_size.py; the reference's own FFT-swap self-noise is 8.9e-5.)"""
    x = O.noise(1, 480000, seed=99)
    sr = 48000.0
    pv_r = O.analyze(x, sr, 2048, 512, 2048)
    out_r, _ = O.synthesize(pv_r, sr, sr / 512, 2048)
    pv_g = fa.analyze(x, sr, 2048, 512, 2048)
    out_g, _ = fa.synthesize(pv_g, sr, sr / 512, 2048)
    rms = float(np.sqrt(np.mean((out_g.astype(np.float64) - out_r.astype(np.float64)) ** 2)))
    print("\n[P3 noise 10 s] composite rms diff=%.3e (reference FFT-swap self-noise: 2.6e-5 @5 s, 8.9e-5 @60 s)" % rms)
    assert rms <= 2e-4            # the documented exception, not the 1e-5 tolerance
    assert rms > 1e-6             # if this ever drops to P2's level, the exception above is out of date: tighten it


def test_nan_flag(fa):
    x = O.noise(1, 8000, seed=3)
    pv = O.analyze(x, 48000.0, 1024, 256, 1024)
    pv[0, 3, 17, 1] = np.nan
    out, flag = fa.synthesize(pv, 48000.0, 48000.0 / 256, 1024)
    assert flag == 1                     # AudioPV.cpp:88-89: warn and carry on
    assert out.shape == (1, pv.shape[1] * 256)


def test_chain_length_invariance(fa):
    """the result must not depend on how frames are cut into chains (flanhip_debug_option: chain length): bit-identical analysis,
    synthesis equal to rounding of the overlap partial sums"""
    x = O.noise(2, 60000, seed=8)
    sr = 48000.0
    for (W, hop, dft) in ((2048, 512, 2048), (1024, 256, 1024), (512, 128, 512), (256, 64, 256), (4096, 1024, 8192), (4096, 1024, 16384)):
        res = []
        for L in (4, 7, 64):
            with fa.debug_options(chain_len=L):
                pv = fa.analyze(x, sr, W, hop, dft)
                out, _ = fa.synthesize(pv, sr, sr / hop, W)
            res.append((pv, out))
        assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
        assert np.array_equal(res[0][0].view(np.uint32), res[2][0].view(np.uint32))
        for k in (1, 2):
            d = np.abs(res[0][1].astype(np.float64) - res[k][1].astype(np.float64))
            assert d.max() <= 2e-6


@pytest.mark.parametrize("W,hop,dft", [(2048, 512, 2048), (2048, 128, 4096), (512, 128, 512), (128, 32, 128), (8192, 2048, 8192)])
def test_many_short_channels_fill_the_kernels_blocks(fa, W, hop, dft):
    """A batch of many short channels (round 6, core.hip choose_chain_length): the cut is made for the kernels' BLOCKS -- groups of 8 wavefronts / 4 teams / 8 ... 32
    chains of ONE channel -- so that few chains per channel do not leave most of a block idle.  600 channels of 0.4 s: the library's own cut against one chain per
    channel and against chains of four frames -- the PV bit for bit, the audio to the rounding of the overlaps' partial sums -- and against the oracle on a few channels."""
    ch, n = 600, 19200
    x = O.noise(ch, n, seed=606)
    sr = 48000.0
    ar = np.float32(sr) / np.float32(hop)
    pv = fa.analyze(x, sr, W, hop, dft)
    out, _ = fa.synthesize(pv, sr, ar, W)
    for L in (1024, 4):                                            # (1024 frames: more than a channel has)
        with fa.debug_options(chain_len=L):
            pv2 = fa.analyze(x, sr, W, hop, dft)
            out2, _ = fa.synthesize(pv, sr, ar, W)
        assert np.array_equal(pv.view(np.uint32), pv2.view(np.uint32)), L
        assert np.abs(out.astype(np.float64) - out2.astype(np.float64)).max() <= 2e-6, L
    sel = [0, 1, 299, 598, 599]
    ref = O.analyze(x[sel], sr, W, hop, dft)
    rel_m, wrms_f, same, turns = p1_metrics(pv[sel], ref, sr / hop)
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    out_sel, _ = fa.synthesize(ref, sr, ar, W)
    rms = float(np.sqrt(np.mean((out_sel.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
    print("\n[600 short channels (%d, %d, %d)] rel_m=%.3e wrms_df=%.3e same=%.4f  P2 rms=%.3e" % (W, hop, dft, rel_m, wrms_f, same, rms))
    assert rel_m <= 1e-5 and wrms_f <= 2e-3 and rms <= 1e-5 and same >= 0.85


@pytest.mark.parametrize("W,hop,dft,n", [(2048, 512, 8192, 150000), (4096, 1024, 16384, 300000), (8192, 2048, 8192, 420000)])
def test_team_kernels_long_chains(fa, W, hop, dft, n):
    """dft 8192 / 16384 (pv_kernels_team.h): chains of more than 64 frames, so that the batches of the k = 512 group (one frame per lane, worked off every
    64 frames) are crossed -- against the oracle, and bit for bit against short chains"""
    x = O.noise(2, n, seed=77)
    sr = 48000.0
    ref = O.analyze(x, sr, W, hop, dft)
    ar = np.float32(sr) / np.float32(hop)
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    res = []
    for L in (150, 65, 9):
        with fa.debug_options(chain_len=L):
            pv = fa.analyze(x, sr, W, hop, dft)
            out, _ = fa.synthesize(ref, sr, ar, W)
        res.append((pv, out))
    rel_m, wrms_f, same, turns = p1_metrics(res[0][0], ref, sr / hop)
    rms = float(np.sqrt(np.mean((res[0][1].astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
    print("\n[team long chains %d %d %d] rel_m=%.3e wrms_df=%.3e same=%.4f  P2 rms=%.3e" % (W, hop, dft, rel_m, wrms_f, same, rms))
    assert rel_m <= 1e-5 and wrms_f <= 5e-4 and same >= 0.985 and rms <= 1e-5
    # the orphan bins by themselves (512 + 1024 j): the same bar
    orph = np.arange(512, dft // 2, 1024)
    rel_o, wrms_o, same_o, _ = p1_metrics(res[0][0][:, :, orph], ref[:, :, orph], sr / hop)
    assert rel_o <= 1e-5 and same_o >= 0.97
    for k in (1, 2):
        assert np.array_equal(res[0][0].view(np.uint32), res[k][0].view(np.uint32))
        assert np.abs(res[0][1].astype(np.float64) - res[k][1].astype(np.float64)).max() <= 2e-6


@pytest.mark.parametrize("W,hop,dft", [(512, 128, 512), (512, 256, 512), (384, 64, 512), (256, 64, 256), (256, 32, 256), (192, 128, 256),
                                       (128, 32, 128), (128, 16, 128), (96, 64, 128)])
def test_sub_kernels_agree_with_their_predecessors(fa, W, hop, dft):
    """dft 512 / 256 / 128 with several chains per wavefront (pv_kernels_sub.h) against the kernels they replaced (FLANHIP_DEBUG_NO_SUB: the one-wavefront kernels of
    pv_kernels_v3.h at dft 512, the generic ones at dft 256 / 128): both within the oracle's tolerances, and of each other; chains of 40 frames so that the batches
    of bin C/2 (one frame per lane of a chain's lane group) are crossed"""
    x = O.noise(3, 100000, seed=5)
    sr = 48000.0
    ar = np.float32(sr) / np.float32(hop)
    ref = O.analyze(x, sr, W, hop, dft)
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    res = {}
    for mode in (0, 1):
        with fa.debug_options(no_sub=mode, chain_len=40):
            pv = fa.analyze(x, sr, W, hop, dft)
            out, _ = fa.synthesize(ref, sr, ar, W)
        rel_m, wrms_f, same, turns = p1_metrics(pv, ref, sr / hop)
        rms = float(np.sqrt(np.mean((out.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
        print("\n[sub kernels off=%d (%d, %d, %d)] rel_m=%.3e wrms_df=%.3e same=%.4f  P2 rms=%.3e" % (mode, W, hop, dft, rel_m, wrms_f, same, rms))
        # (the share of f words bit for bit the oracle's falls with the bin width per analysis rate: 0.927 at (256, 32, 256), 0.888 at (128, 16, 128) -- from either class of kernel)
        assert rel_m <= 1e-5 and wrms_f <= 2e-3 and rms <= 1e-5 and same >= (0.92 if dft >= 256 else 0.85)
        res[mode] = (pv, out, same)
    assert abs(res[0][2] - res[1][2]) <= 0.01                       # the share of f words that are bit for bit the oracle's: the same class of kernel
    assert np.abs(res[0][1].astype(np.float64) - res[1][1].astype(np.float64)).max() <= 5e-6


def test_errors(fa):
    import flan_amd
    x = np.zeros((1, 1000), np.float32)
    with pytest.raises(flan_amd.FlanHipError) as e:
        fa.analyze(x, 48000.0, 2048, 512, 3001)          # an odd dft size (PVBuffer.cpp:356-359 could not represent it: dft = 2 ( bins - 1 ))
    assert e.value.code == flan_amd.ERR_UNSUPPORTED
    assert fa.analyze(x, 48000.0, 2048, 512, 3000).shape == (1, 2, 1501, 2)     # any EVEN size is served, like FFTW behind FFTHelper.cpp:16-26
    with pytest.raises(flan_amd.FlanHipError) as e:
        fa.analyze(x, 48000.0, 4096, 512, 2048)          # window > dft
    assert e.value.code == flan_amd.ERR_INVALID_ARG


@pytest.mark.parametrize("dft,hop", [(2048, 512), (4096, 128), (2048, 1024), (2048, 256), (4096, 1024), (1024, 256), (1024, 1024), (512, 128), (512, 256), (256, 64), (256, 32), (8192, 512), (8192, 1024), (16384, 1024)])
def test_generic_and_tuned_kernels_agree(fa, dft, hop):
    """dft 512 ... 4096 have tuned kernels (pv_kernels_v2.h, _v3.h, _eo.h); the force_generic hook routes the same call through the
    generic ones (pv_kernels.h).  Both must sit within the parity tolerances of the oracle and of each other."""
    x = O.noise(2, 70000, seed=21)
    sr = 48000.0
    W = min(2048, dft) if dft <= 8192 else 4096       # (dft 8192 / 16384: the team kernels against the round-1 block kernels / the mixed-radix kernels)
    ref = O.analyze(x, sr, W, hop, dft)
    out_ref, _ = O.synthesize(ref, sr, np.float32(sr) / np.float32(hop), W)
    res = {}
    for mode in ("0", "1"):
        with fa.debug_options(force_generic=int(mode)):
            pv = fa.analyze(x, sr, W, hop, dft)
            out, _ = fa.synthesize(ref, sr, np.float32(sr) / np.float32(hop), W)
        rel_m, wrms_f, same, turns = p1_metrics(pv, ref, sr / hop)
        rms = float(np.sqrt(np.mean((out.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
        print("\n[path generic=%s dft=%d hop=%d] rel_m=%.3e wrms_df=%.3e same=%.4f turns=%d  P2 rms=%.3e" % (mode, dft, hop, rel_m, wrms_f, same, turns, rms))
        assert rel_m <= 1e-5 and wrms_f <= 2e-3 and rms <= 1e-5
        res[mode] = (pv, out)
    d = np.abs(res["0"][1].astype(np.float64) - res["1"][1].astype(np.float64))
    assert d.max() <= 5e-6


def test_fused_round_trip_equals_unfused(fa):
    """flanhip_analyze_dev_fused + flanhip_synthesize_dev_fused (pre-pass done inside the analysis kernel) must give the very
    same PV and audio as the plain pair, and must carry the NaN flag across."""
    import ctypes
    lib = fa.lib
    sr = 48000.0
    # dft 8192 (and 4096 through the generic kernels): block-wide teams walk the chains; they leave the sums like every other analysis kernel
    for (ch, n, W, hop, dft) in [(2, 70000, 2048, 512, 2048), (1, 30000, 2048, 128, 4096), (2, 20000, 1024, 256, 1024), (1, 9000, 400, 100, 512), (2, 40000, 512, 128, 512), (3, 90000, 1024, 512, 1024),
                                 (3, 300000, 256, 64, 256), (2, 100000, 512, 64, 512), (1, 40000, 4096, 1024, 8192), (2, 400000, 8192, 2048, 8192), (3, 300000, 4096, 1024, 16384), (1, 100000, 4000, 1000, 8192), (2, 30000, 2048, 300, 4096), (3, 500000, 2000, 500, 4096), (2, 200000, 4000, 1000, 4096),
                                 (3, 500000, 1000, 250, 1024), (2, 30000, 1024, 300, 1024), (4, 300000, 500, 125, 512), (1, 20000, 512, 100, 512),
                                 (2, 300000, 2048, 300, 2048), (3, 200000, 2000, 500, 2048), (1, 40000, 1800, 450, 2048), (3, 700000, 3000, 750, 4096), (2, 90000, 4094, 441, 4096),
                                 # the mixed-radix kernels: sums kept by the analysis kernel (ping-pong sizes, with and without the large odd radices) or by
                                 # the pre-pass kernel on its behalf (in place: 12000)
                                 (2, 300000, 2048, 512, 3000), (1, 120000, 1024, 256, 2002), (1, 200000, 2048, 512, 12000), (1, 60000, 600, 150, 1000),
                                 # the chirp-z kernels (sums kept by the ping-pong kernels, with and without the tables in registers; by the pre-pass kernel for the
                                 # in-place layout) and the residue-pair kernels above 16384 (pre-pass kernel on their behalf)
                                 (2, 300000, 2048, 512, 2998), (1, 120000, 1024, 256, 2018), (1, 200000, 2048, 512, 5998), (2, 400000, 4096, 1024, 32768), (2, 300000, 4096, 1024, 20000)]:
        x = O.noise(ch, n, seed=31)
        F = O.num_pv_frames(n, hop)
        bins = dft // 2 + 1
        ar = np.float32(sr) / np.float32(hop)
        hop_s = lib.flanhip_hop_size(sr, ar)

        def dev_alloc(nbytes):
            p = ctypes.c_void_p()
            fa.check(lib.flanhip_malloc(ctypes.byref(p), nbytes))
            return p
        d_x = dev_alloc(x.nbytes)
        fa.check(lib.flanhip_memcpy_h2d(d_x, x.ctypes.data_as(ctypes.c_void_p), x.nbytes, None))
        ws_bytes = lib.flanhip_synthesize_workspace_bytes(ch, F, bins, sr, ar, W)
        res = []
        for fused in (False, True):
            d_pv, d_out, d_ws, d_flag = dev_alloc(ch * F * bins * 8), dev_alloc(ch * F * hop_s * 4), dev_alloc(ws_bytes), dev_alloc(4)
            fa.check(lib.flanhip_memset(d_flag, 0, 4, None))
            if fused:
                fa.check(lib.flanhip_analyze_dev_fused(d_x, ch, n, sr, W, hop, dft, d_pv, d_ws, None))
                fa.check(lib.flanhip_synthesize_dev_fused(d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_flag, None))
            else:
                fa.check(lib.flanhip_analyze_dev(d_x, ch, n, sr, W, hop, dft, d_pv, None))
                fa.check(lib.flanhip_synthesize_dev(d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_flag, None))
            pv = np.empty((ch, F, bins, 2), np.float32); out = np.empty((ch, F * hop_s), np.float32); flag = np.zeros(1, np.int32)
            for host, dev in ((pv, d_pv), (out, d_out), (flag, d_flag)):
                fa.check(lib.flanhip_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), dev, host.nbytes, None))
            fa.check(lib.flanhip_stream_synchronize(None))
            res.append((pv, out, int(flag[0])))
            for p in (d_pv, d_out, d_ws, d_flag):
                lib.flanhip_free(p)
        lib.flanhip_free(d_x)
        assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
        d = np.abs(res[0][1].astype(np.float64) - res[1][1].astype(np.float64))
        print("\n[fused vs unfused dft=%d hop=%d] max audio diff %.3e" % (dft, hop, d.max()))
        assert d.max() <= 1e-6
        assert res[0][2] == res[1][2] == 0


@pytest.mark.parametrize("W,hop,dft", [(2048, 512, 2048), (1024, 256, 1024), (2048, 512, 4096)])
def test_a_workspace_written_by_another_producer_is_reported(fa, W, hop, dft):
    """Which producer filled a synthesis workspace is a host-side note (core.hip); what is IN the workspace carries the producer's epoch, and the
    synthesis kernels that take their carries from the analysis' group totals check the one against the other: a workspace another producer wrote
    in between (two callers racing on one workspace) raises bit 1 of the flag (value 2) instead of passing silently.  Emulated by copying a second
    round trip's workspace over the first's behind the library's back."""
    import ctypes
    lib = fa.lib
    sr = 48000.0
    ch, n = 2, 300000
    F = O.num_pv_frames(n, hop)
    bins = dft // 2 + 1
    ar = np.float32(sr) / np.float32(hop)

    def dev_alloc(nbytes):
        p = ctypes.c_void_p()
        fa.check(lib.flanhip_malloc(ctypes.byref(p), nbytes))
        return p
    ws_bytes = lib.flanhip_synthesize_workspace_bytes(ch, F, bins, sr, ar, W)
    d_x = [dev_alloc(ch * n * 4) for _ in range(2)]
    d_pv = [dev_alloc(ch * F * bins * 8) for _ in range(2)]
    d_ws = [dev_alloc(ws_bytes) for _ in range(2)]
    d_out, d_flag = dev_alloc(ch * F * hop * 4), dev_alloc(4)
    try:
        for i in range(2):
            x = O.noise(ch, n, seed=100 + i)
            fa.check(lib.flanhip_memcpy_h2d(d_x[i], x.ctypes.data_as(ctypes.c_void_p), x.nbytes, None))
            fa.check(lib.flanhip_analyze_dev_fused(d_x[i], ch, n, sr, W, hop, dft, d_pv[i], d_ws[i], None))
        flag = np.zeros(1, np.int32)

        def synth(i):
            fa.check(lib.flanhip_memset(d_flag, 0, 4, None))
            fa.check(lib.flanhip_synthesize_dev_fused(d_pv[i], ch, F, bins, sr, ar, W, d_out, d_ws[i], d_flag, None))
            fa.check(lib.flanhip_memcpy_d2h(flag.ctypes.data_as(ctypes.c_void_p), d_flag, 4, None))
            fa.check(lib.flanhip_stream_synchronize(None))
            return int(flag[0])
        assert synth(0) == 0 and synth(1) == 0 and synth(0) == 0               # the producers' own workspaces, in any order, as often as wanted
        host = np.empty(ws_bytes, np.uint8)
        fa.check(lib.flanhip_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), d_ws[1], ws_bytes, None))
        fa.check(lib.flanhip_stream_synchronize(None))
        fa.check(lib.flanhip_memcpy_h2d(d_ws[0], host.ctypes.data_as(ctypes.c_void_p), ws_bytes, None))
        assert synth(0) & 2                                                    # workspace 0 now holds producer 1's sums: reported
        assert synth(1) == 0
    finally:
        for p in d_x + d_pv + d_ws + [d_out, d_flag]:
            lib.flanhip_free(p)


@pytest.mark.parametrize("dft,hop,ch,n,W", [(2048, 512, 8, 300000, 2048), (4096, 512, 4, 600000, 2048), (4096, 128, 2, 400000, 2048), (4096, 1024, 4, 1400000, 2048),
                                            (2048, 512, 3, 900000, 2048), (4096, 1024, 4, 1400000, 4096), (4096, 512, 4, 600000, 3072), (2048, 512, 1, 2000000, 2048),
                                            (4096, 441, 4, 600000, 2048), (4096, 500, 2, 900000, 2000), (1024, 256, 8, 600000, 1024), (1024, 512, 3, 900000, 768), (1024, 128, 1, 2000000, 1024), (512, 128, 2, 400000, 512), (512, 256, 5, 300000, 512),
                                            (1024, 250, 4, 600000, 1000), (1024, 300, 2, 900000, 1024), (512, 125, 3, 400000, 500),
                                            (2048, 441, 4, 600000, 2048), (2048, 500, 2, 900000, 2000), (4096, 1000, 4, 1400000, 4000), (4096, 441, 2, 900000, 3000)])
def test_carry_prologue_equals_scan_kernel(fa, dft, hop, ch, n, W):
    """Fused round trip: the synthesis kernels that take their chains' carries from a scan over the analysis' GROUP totals plus the chain sums
    (dft 2048 / 1024: groups of 8 chains; dft 4096 team kernels and dft 512: groups of 4, the last group of a channel short; also the team kernels'
    one-buffer-set variants for windows above 2048) against the same launch with the scan over the chains themselves in front (synthesis variant 2); 147-977 chains per
    channel: the same prefix sums in another association -- audio bit for bit but for the rare sample a carry's last bit decides (see below), NaN flag clear."""
    import ctypes
    lib = fa.lib
    sr = 48000.0
    x = O.noise(ch, n, seed=dft + hop)
    F = O.num_pv_frames(n, hop)
    bins = dft // 2 + 1
    ar = np.float32(sr) / np.float32(hop)

    def dev_alloc(nbytes):
        p = ctypes.c_void_p()
        fa.check(lib.flanhip_malloc(ctypes.byref(p), nbytes))
        return p
    d_x = dev_alloc(x.nbytes)
    fa.check(lib.flanhip_memcpy_h2d(d_x, x.ctypes.data_as(ctypes.c_void_p), x.nbytes, None))
    ws_bytes = lib.flanhip_synthesize_workspace_bytes(ch, F, bins, sr, ar, W)
    outs = {}
    try:
        for variant in (0, 2):
            lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, variant)
            d_pv, d_out, d_ws, d_flag = dev_alloc(ch * F * bins * 8), dev_alloc(ch * F * hop * 4), dev_alloc(ws_bytes), dev_alloc(4)
            fa.check(lib.flanhip_memset(d_flag, 0, 4, None))
            fa.check(lib.flanhip_analyze_dev_fused(d_x, ch, n, sr, W, hop, dft, d_pv, d_ws, None))
            fa.check(lib.flanhip_synthesize_dev_fused(d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_flag, None))
            out = np.empty((ch, F * hop), np.float32); flag = np.zeros(1, np.int32)
            fa.check(lib.flanhip_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), d_out, out.nbytes, None))
            fa.check(lib.flanhip_memcpy_d2h(flag.ctypes.data_as(ctypes.c_void_p), d_flag, 4, None))
            fa.check(lib.flanhip_stream_synchronize(None))
            outs[variant] = (out, int(flag[0]))
            for p in (d_pv, d_out, d_ws, d_flag):
                lib.flanhip_free(p)
    finally:
        lib.flanhip_debug_option(fa.DEBUG_SYN_VARIANT, 0)
        lib.flanhip_free(d_x)
    assert outs[0][1] == 0 and outs[2][1] == 0
    # The two forms add the same sums in two associations (groups of 8 / 4 chains, then the groups; against segments of 16 - 64 chains): carries
    # equal to ~1e-15, and a carry's last bits decide a sample only where some frame's phase lies that close to a rounding boundary of its float
    # -- no chain in most launches, one or two in a few (tools/carry_forms_agree.py: 3 of 96 shape x seed pairs, 22 - 151 samples of 1.2 - 2.4 M, one
    # ulp each).  So: bit for bit but for a few samples in a few chains, those by one ulp of the signal's scale.
    differing = int((outs[0][0].view(np.uint32) != outs[2][0].view(np.uint32)).sum())
    assert differing <= 4e-4 * outs[0][0].size
    assert np.abs(outs[0][0].astype(np.float64) - outs[2][0].astype(np.float64)).max() <= 2.4e-7
    assert np.abs(outs[0][0]).max() > 0.1


def _random_shapes(count, seed):
    """seeded shapes across every kernel variant: tuned dft 2048 / 4096 with hops that do / do not suit the register overlap-add,
    generic dft 32 ... 8192, windows shorter than the dft, odd windows and hops, hops beyond the window, inputs shorter than a window"""
    rng = np.random.default_rng(seed)
    shapes = []
    for i in range(count):
        dft = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048, 2048, 2048, 4096, 4096, 8192]))
        W = dft if rng.random() < 0.4 else int(rng.integers(max(dft // 8, 2), dft + 1))
        if dft >= 2048 and rng.random() < 0.5:
            hop = int(rng.choice([128, 256, 384, 512, 640, 1024]))
        else:
            hop = int(rng.integers(1 if dft <= 256 else max(dft // 64, 1), 2 * W))
        ch = int(rng.integers(1, 4))
        frames = int(rng.integers(1, 40))
        n = max(int(frames * hop + rng.integers(-hop // 2, hop // 2 + 1)), 1)
        shapes.append((ch, n, W, hop, dft))
    return shapes


@pytest.mark.parametrize("ch,n,W,hop,dft", _random_shapes(36, 20260101), ids=lambda v: str(v))
def test_random_shapes(fa, ch, n, W, hop, dft):
    sr = 48000.0
    x = O.noise(ch, n, seed=ch * 7 + W + hop)
    ref = O.analyze(x, sr, W, hop, dft)
    got = fa.analyze(x, sr, W, hop, dft)
    assert got.shape == ref.shape
    ar = np.float32(sr) / np.float32(hop)
    rel_m, wrms_f, same, turns = p1_metrics(got, ref, float(ar))
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    if O.lib.oracle_hop_size(sr, ar) != hop:
        pytest.skip("hop %d is not recovered from sample_rate / analysis_rate in fp32 (PVBuffer.cpp:381-384): synthesis undefined" % hop)
    out_got, flag = fa.synthesize(ref, sr, ar, W)
    assert out_got.shape == out_ref.shape and flag == 0
    scale = max(float(np.sqrt(np.mean(out_ref.astype(np.float64) ** 2))), 1e-30)
    rms = float(np.sqrt(np.mean((out_got.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
    print("\n[random %s] P1 rel_m=%.2e wrms_df=%.2e same=%.4f  P2 rms=%.2e (signal rms %.2e)" % ((ch, n, W, hop, dft), rel_m, wrms_f, same, rms, scale))
    assert rel_m <= 1e-5
    assert wrms_f <= 2e-3 * max(sr / dft / 23.4, 1.0)
    assert rms <= 1e-5 * max(scale, 1.0)


def _random_round6_shapes(count, seed):
    """seeded shapes ON the grids of round 6's kernel families (pv_kernels_sub.h: dft 512 / 256, windows and hops multiples of 64 / 32 samples, hop 1 / 2 / 4 / 8
    steps; pv_kernels_team.h: dft 8192 / 16384, windows of 4 / 8 / 16 steps of 512 / 1024 samples): ragged lengths from less than a hop to a few hundred frames,
    1 - 6 channels, and a forced chain length in a third of them (short last chains, spare lane groups, chains of one frame)"""
    rng = np.random.default_rng(seed)
    shapes = []
    for i in range(count):
        dft = int(rng.choice([256, 512, 512, 8192, 16384]))
        if dft <= 512:
            step = dft // 8
            hop = step * int(rng.choice([1, 2, 4, 8]))
            W = step * int(rng.integers(max(hop // step, 1), 9))
        else:
            step = dft // 16
            wq = int(rng.choice([4, 8, 16]))
            hs = int(rng.choice([h for h in (1, 2, 4, 8) if h <= wq and (wq, h) in ((4, 1), (4, 2), (8, 1), (8, 2), (8, 4), (16, 2), (16, 4), (16, 8))]))
            W, hop = wq * step, hs * step
        ch = int(rng.integers(1, 7))
        frames = int(rng.choice([1, 2, 3, 5, 17, 40, 90, 300])) if dft <= 512 else int(rng.choice([1, 2, 3, 9, 30, 70]))
        n = max(int(frames * hop + rng.integers(-hop // 2, hop // 2 + 1)), 2)
        chain_len = int(rng.choice([0, 0, 1, 3, 8, 33]))
        shapes.append((ch, n, W, hop, dft, chain_len))
    return shapes


@pytest.mark.parametrize("ch,n,W,hop,dft,chain_len", _random_round6_shapes(40, 20261006), ids=lambda v: str(v))
def test_random_shapes_of_the_round6_families(fa, ch, n, W, hop, dft, chain_len):
    sr = 48000.0
    x = O.noise(ch, n, seed=ch * 11 + W + hop)
    ref = O.analyze(x, sr, W, hop, dft)
    ar = np.float32(sr) / np.float32(hop)
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    with fa.debug_options(chain_len=chain_len):
        got = fa.analyze(x, sr, W, hop, dft)
        out_got, flag = fa.synthesize(ref, sr, ar, W)
    assert got.shape == ref.shape and out_got.shape == out_ref.shape and flag == 0
    rel_m, wrms_f, same, turns = p1_metrics(got, ref, float(ar))
    scale = max(float(np.sqrt(np.mean(out_ref.astype(np.float64) ** 2))), 1e-30)
    rms = float(np.sqrt(np.mean((out_got.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
    print("\n[round-6 random %s] P1 rel_m=%.2e wrms_df=%.2e same=%.4f  P2 rms=%.2e (signal rms %.2e)" % ((ch, n, W, hop, dft, chain_len), rel_m, wrms_f, same, rms, scale))
    assert rel_m <= 1e-5 and wrms_f <= 2e-3 * max(sr / dft / 23.4, 1.0) and rms <= 1e-5 * max(scale, 1.0)
    assert same >= 0.90


def _random_smooth_shapes(count, seed):
    """seeded shapes for the mixed-radix kernels (pv_kernels_mr.h): dft = 2 C with C a product of 2, 3, 5, 7, 11, 13 (not a power of two), ping-pong
    and in-place sizes, plans with and without the large odd radices; windows up to the dft, hops up to beyond the window, ragged lengths"""
    rng = np.random.default_rng(seed)
    smooth = sorted({2 * a * b for a in (1, 2, 4, 8, 16, 32, 64, 128, 256) for b in (3, 5, 7, 9, 11, 13, 15, 21, 25, 27, 33, 35, 39, 45, 55, 63, 75, 77, 91, 125, 143, 375)
                     if 24 <= 2 * a * b <= 16384})
    shapes = []
    for i in range(count):
        dft = int(rng.choice(smooth))
        W = dft if rng.random() < 0.4 else int(rng.integers(max(dft // 8, 2), dft + 1))
        hop = int(rng.integers(max(dft // 64, 1), 2 * W)) if rng.random() < 0.3 else int(rng.integers(max(W // 8, 1), W + 1))
        ch = int(rng.integers(1, 4))
        frames = int(rng.integers(1, 30))
        n = max(int(frames * hop + rng.integers(-hop // 2, hop // 2 + 1)), 1)
        shapes.append((ch, n, W, hop, dft))
    return shapes


@pytest.mark.parametrize("ch,n,W,hop,dft", _random_smooth_shapes(28, 20261004), ids=lambda v: str(v))
def test_random_smooth_sizes(fa, ch, n, W, hop, dft):
    sr = 48000.0
    x = O.noise(ch, n, seed=ch * 11 + W + hop)
    ref = O.analyze(x, sr, W, hop, dft)
    got = fa.analyze(x, sr, W, hop, dft)
    assert got.shape == ref.shape and np.all(np.isfinite(got))
    ar = np.float32(sr) / np.float32(hop)
    rel_m, wrms_f, same, turns = p1_metrics(got, ref, float(ar))
    if O.lib.oracle_hop_size(sr, ar) != hop:
        pytest.skip("hop %d is not recovered from sample_rate / analysis_rate in fp32 (PVBuffer.cpp:381-384): synthesis undefined" % hop)
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    out_got, flag = fa.synthesize(ref, sr, ar, W)
    assert out_got.shape == out_ref.shape and flag == 0
    scale = max(float(np.sqrt(np.mean(out_ref.astype(np.float64) ** 2))), 1e-30)
    rms = float(np.sqrt(np.mean((out_got.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
    print("\n[smooth %s] P1 rel_m=%.2e wrms_df=%.2e same=%.4f  P2 rms=%.2e (signal rms %.2e)" % ((ch, n, W, hop, dft), rel_m, wrms_f, same, rms, scale))
    assert rel_m <= 1e-5
    assert wrms_f <= 2e-3 * max(sr / dft / 23.4, 1.0)
    assert rms <= 1e-5 * max(scale, 1.0)


def _random_chirp_and_big_shapes(count, seed):
    """seeded shapes for the chirp-z kernels (pv_kernels_bs.h: half the size with a prime factor above 13, both layouts, M = 128 ... 8192) and for the
    residue-pair kernels above 16384 (pv_kernels_big.h: C1 = 2 ... 9 and 32, C2 = 1024 ... 4096, windows of one to three segments); windows up to the
    dft, hops up to beyond the window, ragged lengths, one to three channels"""
    rng = np.random.default_rng(seed)
    primes = [67, 101, 131, 257, 331, 509, 521, 769, 1009, 1031, 1499, 2039, 2053, 2999, 4093]
    chirp = sorted({2 * p * a for p in primes for a in (1, 2, 3, 4, 6) if 64 <= p * a <= 4096})
    big = [2 * 1024 * c1 for c1 in (9, 10, 11, 14, 18)] + [2 * 2048 * c1 for c1 in (5, 7, 9)] + [2 * 4096 * c1 for c1 in (2, 3, 4, 5, 6, 8, 32)]
    shapes = []
    for i in range(count):
        is_big = i % 3 == 2
        dft = int(rng.choice(big if is_big else chirp))
        if is_big:
            W = int(rng.choice([1024, 2048, 4096, 5000, 8192, 12000]))
            W = min(W, dft, (2 ** 31 - 1) // dft)      # (dft x window beyond int32 is refused: the reference's own product overflows, AudioPV.cpp:99)
        else:
            W = dft if rng.random() < 0.4 else int(rng.integers(max(dft // 8, 2), dft + 1))
        hop = int(rng.integers(max(W // 16, 1), 2 * W)) if rng.random() < 0.25 else int(rng.integers(max(W // 8, 1), W + 1))
        ch = int(rng.integers(1, 4))
        frames = int(rng.integers(1, 12 if is_big else 30))
        n = max(int(frames * hop + rng.integers(-hop // 2, hop // 2 + 1)), 1)
        shapes.append((ch, n, W, hop, dft))
    return shapes


@pytest.mark.parametrize("ch,n,W,hop,dft", _random_chirp_and_big_shapes(30, 20261005), ids=lambda v: str(v))
def test_random_chirp_z_and_big_sizes(fa, ch, n, W, hop, dft):
    sr = 48000.0
    x = O.noise(ch, n, seed=ch * 13 + W + hop)
    ref = O.analyze(x, sr, W, hop, dft)
    got = fa.analyze(x, sr, W, hop, dft)
    assert got.shape == ref.shape and np.all(np.isfinite(got))
    ar = np.float32(sr) / np.float32(hop)
    rel_m, wrms_f, same, turns = p1_metrics(got, ref, float(ar))
    if O.lib.oracle_hop_size(sr, ar) != hop:
        pytest.skip("hop %d is not recovered from sample_rate / analysis_rate in fp32 (PVBuffer.cpp:381-384): synthesis undefined" % hop)
    out_ref, _ = O.synthesize(ref, sr, ar, W)
    out_got, flag = fa.synthesize(ref, sr, ar, W)
    assert out_got.shape == out_ref.shape and flag == 0
    scale = max(float(np.sqrt(np.mean(out_ref.astype(np.float64) ** 2))), 1e-30)
    rms = float(np.sqrt(np.mean((out_got.astype(np.float64) - out_ref.astype(np.float64)) ** 2)))
    print("\n[chirp / big %s] P1 rel_m=%.2e wrms_df=%.2e same=%.4f  P2 rms=%.2e (signal rms %.2e)" % ((ch, n, W, hop, dft), rel_m, wrms_f, same, rms, scale))
    assert rel_m <= 1e-5
    assert wrms_f <= 2e-3 * max(sr / dft / 23.4, 1.0)
    assert rms <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize("n", [0, 1, 2, 3])
def test_degenerate_lengths(fa, n):
    """empty and near-empty signals: one frame of (almost) silence through every dft class, like the reference would produce"""
    sr = 48000.0
    for (W, hop, dft) in ((2048, 512, 2048), (2048, 128, 4096), (256, 64, 256), (1024, 256, 1024), (512, 128, 512), (1024, 256, 2998), (4096, 1024, 32768), (4096, 1024, 20000), (8192, 2048, 8192)):
        x = O.noise(2, max(n, 1), seed=5)[:, :n].copy()
        ref = O.analyze(x, sr, W, hop, dft)
        got = fa.analyze(x, sr, W, hop, dft)
        assert got.shape == ref.shape == (2, 1, dft // 2 + 1, 2)
        m_r = ref[..., 0].astype(np.float64)
        assert np.abs(got[..., 0] - m_r).max() <= 1e-6 * max(m_r.max(), 1e-30) + 1e-12
        ar = np.float32(sr) / np.float32(hop)
        out_ref, _ = O.synthesize(ref, sr, ar, W)
        out_got, flag = fa.synthesize(ref, sr, ar, W)
        assert out_got.shape == out_ref.shape and flag == 0
        assert np.abs(out_got - out_ref).max() <= 1e-6


def _special_signals(n):
    t = np.arange(n)
    sigs = {
        "impulse": np.where(t == 700, 1.0, 0.0),
        "impulse-train": np.where(t % 512 == 0, 1.0, 0.0),
        "dc": np.full(n, 0.25),
        "nyquist": np.where(t % 2 == 0, 0.5, -0.5),
        "square": np.where((t // 50) % 2 == 0, 1.0, -1.0),
        "bin-centred-sine": np.sin(2 * np.pi * (48000.0 * 40 / 2048) * t / 48000.0),       # lands exactly on bin 40: exact zeros elsewhere (up to rounding)
        "tiny-noise": O.noise(1, n, seed=1)[0] * 1e-30,
        "huge-noise": O.noise(1, n, seed=2)[0] * 1e30,
        "denormal-noise": O.noise(1, n, seed=3)[0] * 1e-42,
        "step": np.where(t > n // 2, 1.0, 0.0),
    }
    return {k: np.ascontiguousarray(v, np.float32)[None, :] for k, v in sigs.items()}


@pytest.mark.parametrize("dft,hop", [(2048, 512), (4096, 128), (1024, 256)])
def test_special_signals(fa, dft, hop):
    """signals that put exact zeros, exact axis angles, denormals and very large values through the per-bin math"""
    sr = 48000.0
    W = min(dft, 2048)
    ar = np.float32(sr) / np.float32(hop)
    for name, x in _special_signals(9000).items():
        ref = O.analyze(x, sr, W, hop, dft)
        got = fa.analyze(x, sr, W, hop, dft)
        m_r, m_g = ref[..., 0].astype(np.float64), got[..., 0].astype(np.float64)
        scale = max(m_r.max(), 1e-300)
        rel = np.sqrt(np.sum((m_g - m_r) ** 2)) / max(np.sqrt(np.sum(m_r ** 2)), 1e-300)
        assert np.isfinite(got).all(), name
        # bins that hold rounding noise of an exactly cancelling sum have no meaningful relative error: judge against the frame's scale
        # (denormal spectra: fp32 butterflies keep only the denormals' few bits, the oracle's fp64 FFT keeps all -- absolute floor)
        assert rel <= 1e-5 or np.abs(m_g - m_r).max() <= max(2e-7 * scale, 1e-43), (name, rel)
        if scale < 1e-35:
            continue
        # f of frame t hangs on the phase of frame t - 1 as well: both must stand clear of the rounding noise of the transform
        sig = m_r > 1e-3 * scale
        sig[:, 1:] &= m_r[:, :-1] > 1e-3 * scale
        df = np.abs(got[..., 1].astype(np.float64) - ref[..., 1])[sig]
        turns = np.rint(df / float(ar))
        assert (np.abs(df - turns * float(ar)) <= 0.05).all(), (name, df.max() if df.size else 0)
        out_ref, _ = O.synthesize(ref, sr, ar, W)
        out_got, flag = fa.synthesize(ref, sr, ar, W)
        s_out = max(float(np.abs(out_ref).max()), 1e-300)
        assert flag == 0 and np.isfinite(out_got).all(), name
        assert np.abs(out_got.astype(np.float64) - out_ref).max() <= 2e-5 * s_out + 1e-37, (name, np.abs(out_got.astype(np.float64) - out_ref).max(), s_out)


@pytest.mark.parametrize("hop", [128, 256, 512, 1024])
def test_dft4096_kernel_generations_agree(fa, hop):
    """dft 4096, window 2048: the team kernels (two 1024-point register transforms per frame: pv_kernels_eo.h) against their A/B predecessor, the generic
    block-per-chain kernels of pv_kernels.h (FLANHIP_DEBUG_ANA4096_OLD / SYN4096_OLD; until round 6 these hooks selected round 1's own tuned kernels, now
    retired), on the same input: the same per-bin arithmetic behind two FFT factorisations -- PVs agree like two FFT backends do (magnitudes 1e-7, most f bit
    for bit), audio from the SAME PV to 1e-6."""
    sr, W, dft = 48000.0, 2048, 4096
    x = O.noise(2, 60000 + 7 * hop, seed=hop)
    ar = np.float32(sr) / np.float32(hop)
    res = {}
    try:
        for gen in (0, 1):
            fa.lib.flanhip_debug_option(fa.DEBUG_ANA4096_OLD, 1 - gen)
            fa.lib.flanhip_debug_option(fa.DEBUG_SYN4096_OLD, 1 - gen)
            pv = fa.analyze(x, sr, W, hop, dft)
            res[gen] = pv
    finally:
        fa.lib.flanhip_debug_option(fa.DEBUG_ANA4096_OLD, 0)
        fa.lib.flanhip_debug_option(fa.DEBUG_SYN4096_OLD, 0)
    m0, m1 = res[0][..., 0].astype(np.float64), res[1][..., 0].astype(np.float64)
    rel_m = np.sqrt(np.sum((m0 - m1) ** 2) / np.sum(m0 ** 2))
    same_f = np.mean(res[0][..., 1].view(np.uint32) == res[1][..., 1].view(np.uint32))
    out = {}
    try:
        for gen in (0, 1):
            fa.lib.flanhip_debug_option(fa.DEBUG_SYN4096_OLD, 1 - gen)
            out[gen], flag = fa.synthesize(res[1], sr, ar, W)
            assert flag == 0
    finally:
        fa.lib.flanhip_debug_option(fa.DEBUG_SYN4096_OLD, 0)
    d = np.abs(out[0].astype(np.float64) - out[1].astype(np.float64))
    print("\n[dft 4096 hop %d] generations: rel_m %.2e  f bit-identical %.4f  audio max diff %.2e" % (hop, rel_m, same_f, d.max()))
    assert rel_m <= 5e-7 and same_f >= 0.95 and d.max() <= 1e-6


def test_cancellation_inside_a_launch(fa):
    """defines.h:49-62 / AudioPV.cpp:49,115: the reference polls its canceller once per frame.  Here a flag raised WHILE the kernels run stops the
    launch (flanhip_wait_cancellable: the thread's cancel word, read by every block when it starts and by the direct-sum kernels every few
    batches): the wait returns FLANHIP_ERR_CANCELLED long before the work would have completed, and the thread's next call is unharmed."""
    import ctypes
    import threading
    import time
    import torch
    import flan_amd
    lib, vp = fa.lib, ctypes.c_void_p
    dev = torch.device("cuda", 0)
    fa.check(lib.flanhip_set_device(0))
    sr, W, hop, dft = 48000.0, 4096, 1024, 9998                   # a direct-sum size (2 x 4999, a prime): a tenth of a second of kernel time
    ch, n = 8, 90 * 48000
    F = int(lib.flanhip_num_pv_frames(n, hop))
    x = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(vp(x.data_ptr()), ch, n, 5, None))
    pv = torch.empty((ch, F, dft // 2 + 1, 2), dtype=torch.float32, device=dev)

    def launch():
        fa.check(lib.flanhip_analyze_dev(vp(x.data_ptr()), ch, n, sr, W, hop, dft, vp(pv.data_ptr()), None))

    flag = ctypes.c_int(0)
    launch()
    t0 = time.perf_counter()
    assert lib.flanhip_wait_cancellable(None, ctypes.byref(flag)) == 0            # nobody raises the flag: an ordinary wait
    launch()
    t0 = time.perf_counter()
    assert lib.flanhip_wait_cancellable(None, ctypes.byref(flag)) == 0
    full = time.perf_counter() - t0
    ref = pv[:, :3].clone()

    def raise_later():
        time.sleep(0.15 * full)
        flag.value = 1
    th = threading.Thread(target=raise_later)
    pv.zero_()
    launch()
    t0 = time.perf_counter()
    th.start()
    rc = lib.flanhip_wait_cancellable(None, ctypes.byref(flag))
    cut = time.perf_counter() - t0
    th.join()
    print("\n[cancel] full launch %.1f ms, cancelled after %.1f ms (flag raised at %.1f ms)" % (full * 1e3, cut * 1e3, 0.15 * full * 1e3))
    assert rc == flan_amd.ERR_CANCELLED
    assert cut <= 0.6 * full                                                          # it did not run to its end
    flag.value = 0
    launch()                                                                          # the thread's word was reset: the next call is whole
    assert lib.flanhip_wait_cancellable(None, ctypes.byref(flag)) == 0
    assert torch.equal(pv[:, :3].view(torch.int32), ref.view(torch.int32))
    # the FFT kernels stop at block granularity: a launch of SEVERAL rounds of blocks (chains are capped at 512 frames and 2048 of them are
    # resident: 16 ch x 600 s at hop 128 = 3.6 M frames is four rounds, ~12 ms) loses the rounds that have not started
    del x, pv
    W2, hop2, dft2, ch2 = 2048, 128, 2048, 16
    n2 = 600 * 48000
    F2 = int(lib.flanhip_num_pv_frames(n2, hop2))
    x2 = torch.empty((ch2, n2), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(vp(x2.data_ptr()), ch2, n2, 6, None))
    pv2 = torch.empty((ch2, F2, dft2 // 2 + 1, 2), dtype=torch.float32, device=dev)
    go = lambda: fa.check(lib.flanhip_analyze_dev(vp(x2.data_ptr()), ch2, n2, sr, W2, hop2, dft2, vp(pv2.data_ptr()), None))
    go()
    assert lib.flanhip_wait_cancellable(None, ctypes.byref(flag)) == 0
    go()
    t0 = time.perf_counter()
    assert lib.flanhip_wait_cancellable(None, ctypes.byref(flag)) == 0
    full2 = time.perf_counter() - t0
    th = threading.Thread(target=lambda: (time.sleep(0.1 * full2), setattr(flag, "value", 1)))
    go()
    t0 = time.perf_counter()
    th.start()
    rc = lib.flanhip_wait_cancellable(None, ctypes.byref(flag))
    cut2 = time.perf_counter() - t0
    th.join()
    print("[cancel, FFT kernels] full launch %.1f ms, cancelled after %.1f ms" % (full2 * 1e3, cut2 * 1e3))
    assert rc == flan_amd.ERR_CANCELLED and cut2 <= 0.8 * full2


def test_cancellation_is_scoped_to_the_stream_waited_on(fa):
    """A thread with conversions in flight on TWO streams cancels the wait on one of them: the other stream's kernels run to completion and
    its later plain synchronisation reports a whole result (round 3 kept one cancel word per thread: the second stream's kernels stopped
    starting chains too, and its synchronisation returned FLANHIP_OK on a half-written PV)."""
    import ctypes
    import threading
    import time
    import torch
    import flan_amd
    lib, vp = fa.lib, ctypes.c_void_p
    dev = torch.device("cuda", 0)
    fa.check(lib.flanhip_set_device(0))
    sA, sB = vp(), vp()
    fa.check(lib.flanhip_stream_create(ctypes.byref(sA)))
    fa.check(lib.flanhip_stream_create(ctypes.byref(sB)))
    sr, W, hop, dft = 48000.0, 2048, 128, 2048
    ch, n = 16, 600 * 48000                                        # four rounds of blocks: ~10 ms
    F = int(lib.flanhip_num_pv_frames(n, hop))
    x = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(vp(x.data_ptr()), ch, n, 6, None))
    pvA = torch.empty((ch, F, dft // 2 + 1, 2), dtype=torch.float32, device=dev)
    chB, nB = 2, 300 * 48000
    FB = int(lib.flanhip_num_pv_frames(nB, hop))
    pvB = torch.empty((chB, FB, dft // 2 + 1, 2), dtype=torch.float32, device=dev)
    refB = torch.empty_like(pvB)
    fa.check(lib.flanhip_analyze_dev(vp(x.data_ptr()), chB, nB, sr, W, hop, dft, vp(refB.data_ptr()), None))
    torch.cuda.synchronize()
    flag = ctypes.c_int(0)
    goA = lambda: fa.check(lib.flanhip_analyze_dev(vp(x.data_ptr()), ch, n, sr, W, hop, dft, vp(pvA.data_ptr()), sA))
    goA()
    t0 = time.perf_counter()
    assert lib.flanhip_wait_cancellable(sA, ctypes.byref(flag)) == 0
    goA()
    t0 = time.perf_counter()
    assert lib.flanhip_wait_cancellable(sA, ctypes.byref(flag)) == 0
    full = time.perf_counter() - t0
    pvB.zero_()
    torch.cuda.synchronize()
    th = threading.Thread(target=lambda: (time.sleep(0.1 * full), setattr(flag, "value", 1)))
    goA()
    fa.check(lib.flanhip_analyze_dev(vp(x.data_ptr()), chB, nB, sr, W, hop, dft, vp(pvB.data_ptr()), sB))   # the same thread, another stream
    th.start()
    rc = lib.flanhip_wait_cancellable(sA, ctypes.byref(flag))
    th.join()
    assert rc == flan_amd.ERR_CANCELLED
    fa.check(lib.flanhip_stream_synchronize(sB))
    torch.cuda.synchronize()
    assert torch.equal(pvB.view(torch.int32), refB.view(torch.int32)), "the cancellation of stream A reached the kernels of stream B"
    flag.value = 0
    fa.check(lib.flanhip_stream_destroy(sA))
    fa.check(lib.flanhip_stream_destroy(sB))


@pytest.mark.parametrize("W,hop,dft", [(2048, 512, 3000), (4096, 1024, 16384), (600, 150, 1000), (1024, 256, 2998), (2048, 512, 5998), (4096, 1024, 32768),
                                       (4096, 1024, 20000), (2048, 512, 22050), (2048, 512, 17836), (8192, 256, 16384), (10000, 2500, 16384),
                                       (4096, 1024, 9998), (2000, 500, 10002), (8000, 2000, 30002), (4096, 1024, 262142)])   # (262142 = 2 x ( 2^17 - 1 ): the largest chirp-z size, M = 2^18)
def test_mixed_radix_kernels_against_the_direct_sums(fa, W, hop, dft):
    """The mixed-radix FFT kernels (pv_kernels_mr.h) against the transform's definition summed in fp64 (pv_kernels_any.h, the force_direct hook)
    on the same input: PVs agree like two FFT backends do, audio from the SAME PV to 1e-6."""
    sr = 48000.0
    x = O.noise(2, 40000 + 3 * hop, seed=dft)
    ar = np.float32(sr) / np.float32(hop)
    pv_fft = fa.analyze(x, sr, W, hop, dft)
    out_fft, _ = fa.synthesize(pv_fft, sr, ar, W)
    with fa.debug_options(force_direct=1):
        pv_def = fa.analyze(x, sr, W, hop, dft)
        out_def, _ = fa.synthesize(pv_fft, sr, ar, W)
    m0, m1 = pv_def[..., 0].astype(np.float64), pv_fft[..., 0].astype(np.float64)
    rel_m = np.sqrt(np.sum((m0 - m1) ** 2) / np.sum(m0 ** 2))
    same_f = np.mean(pv_def[..., 1].view(np.uint32) == pv_fft[..., 1].view(np.uint32))
    d = np.abs(out_fft.astype(np.float64) - out_def.astype(np.float64)).max()
    print("\n[dft %d] FFT kernels vs direct sums: rel_m %.2e  f bit-identical %.4f  audio max diff %.2e" % (dft, rel_m, same_f, d))
    assert rel_m <= 5e-7 and same_f >= 0.95 and d <= 1e-6


def _big_mixed_plan(dft):
    """bs_plan.h: big_make_plan's choice for a size above 16384 -- (C1, C2, mixed) or None (the host arithmetic restated: tests/cpp/plans_test.cpp holds the original)"""
    def smooth(c):
        for r in (2, 3, 5, 7, 11, 13):
            while c % r == 0:
                c //= r
        return c == 1
    C = dft // 2
    c2 = 1
    while C % (c2 * 2) == 0 and c2 * 2 <= 4096:
        c2 *= 2
    if c2 >= 1024 and C // c2 <= 256:
        return (C // c2, c2, False) if C // c2 >= 2 else None
    for d in range(4096, 255, -1):
        if C % d == 0 and 2 <= C // d <= 256 and smooth(d):
            return (C // d, d, True)
    return None


def test_random_mixed_radix_sizes_above_16384_against_the_direct_sums(fa):
    """Round 6: the residue-pair kernels with a mixed-radix inner transform (bs_plan.h `mixed`), twelve seeded random sizes between 16386 and 300000 that the plan
    serves that way, random windows (one to three segments) and hops, against the direct sums on the same input -- incl. odd C1 / C2 and every odd radix."""
    rng = np.random.default_rng(606)
    sr = 48000.0
    done, radices = 0, set()
    while done < 12:
        dft = 2 * int(rng.integers(8193, 150000))
        plan = _big_mixed_plan(dft)
        if plan is None or not plan[2]:
            continue
        C1, C2, _ = plan
        W = int(rng.integers(64, min(dft, 3 * 2 * C2, 12000)))
        hop = max(1, W // int(rng.integers(2, 9)))
        for r in (3, 5, 7, 11, 13):
            if C2 % r == 0:
                radices.add(r)
        x = O.noise(1 + done % 2, 6 * W + int(rng.integers(0, 999)), seed=dft)
        ar = np.float32(sr) / np.float32(hop)
        pv_fft = fa.analyze(x, sr, W, hop, dft)
        out_fft, _ = fa.synthesize(pv_fft, sr, ar, W)
        with fa.debug_options(force_direct=1):
            pv_def = fa.analyze(x, sr, W, hop, dft)
            out_def, _ = fa.synthesize(pv_fft, sr, ar, W)
        m0, m1 = pv_def[..., 0].astype(np.float64), pv_fft[..., 0].astype(np.float64)
        rel_m = np.sqrt(np.sum((m0 - m1) ** 2) / np.sum(m0 ** 2))
        same_f = np.mean(pv_def[..., 1].view(np.uint32) == pv_fft[..., 1].view(np.uint32))
        d = np.abs(out_fft.astype(np.float64) - out_def.astype(np.float64)).max()
        print("\n[dft %d = 2 x %d x %d, W %d, hop %d] rel_m %.2e  f bit-identical %.4f  audio max diff %.2e" % (dft, C1, C2, W, hop, rel_m, same_f, d))
        assert rel_m <= 5e-7 and same_f >= 0.95 and d <= 1e-6, (dft, W, hop)
        done += 1
    assert radices >= {3, 5, 7}, radices


def test_plan_caches_are_bounded_and_evicted_sizes_come_back(fa):
    """The plan / unit-circle caches keep only the most recently used few of the sizes without tuned kernels (core.hip get_plan, conversions.hip
    get_unit_circle): a sweep over more sizes than they hold, mixed-radix and direct-sum, then the first sizes again -- the results of a size whose
    tables were dropped and rebuilt are bit-identical to its first results, and the device's free memory does not shrink with the sweep."""
    import torch
    sr = 48000.0
    sizes = [96 + 2 * i for i in range(14)] + [2 * 521, 2 * 523, 2 * 541, 2 * 547, 2 * 557, 2 * 563]     # 14 smooth-ish sizes, 6 with a large prime
    first = {}
    def run(dft):
        x = O.noise(1, 20 * dft, seed=dft)
        pv = fa.analyze(x, sr, dft, dft // 4, dft)
        out, _ = fa.synthesize(pv, sr, np.float32(sr) / np.float32(dft // 4), dft)
        return pv, out
    for dft in sizes:
        first[dft] = run(dft)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        for dft in sizes:
            pv, out = run(dft)
            assert np.array_equal(pv.view(np.uint32), first[dft][0].view(np.uint32)) and np.array_equal(out.view(np.uint32), first[dft][1].view(np.uint32)), dft
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 <= (2 << 20), (free0, free1)


@pytest.mark.parametrize("ch,seconds,W,hop,hooks", [
    (8, 60, 2048, 512, {}),                                   # the bench shape: one round of 2048 chains
    (8, 60, 2048, 512, {"chain_len": 7}),                     # 6432 chains on 2048 slots: chains of one boundary run in different rounds
    (2, 60, 2048, 128, {}),                                   # stereo, hop 128 (15 steps per boundary), the scan kernel's layout
    (3, 20, 1024, 256, {"target_chains": 4096}),              # a short window, twice the resident chains
    (5, 33, 2048, 1024, {}),
    (2, 10, 2048, 512, {"chain_len": 3}),                     # chains too short to publish their heads from inside the loop: the exchange at their ends
    (2, 10, 2048, 512, {"chain_len": 4}),
    (3, 7, 2048, 512, {"chain_len": 5}),                      # the head's tag goes out in the last frame but one
    (1, 30, 2048, 256, {"chain_len": 9}),
    # the dft 1024 / 512 kernels (pv_kernels_v3.h): the same protocol
    (8, 60, 1024, 256, {"dft": 1024}),
    (2, 20, 1024, 128, {"dft": 1024, "chain_len": 11}),
    (3, 9, 768, 512, {"dft": 1024, "chain_len": 3}),
    (8, 30, 512, 128, {"dft": 512}),
    (2, 9, 512, 256, {"dft": 512, "chain_len": 4}),
    # the dft 4096 team kernel (pv_kernels_eo.h, round 6): a word per wavefront of a chain; hop 128 (half steps), 256 .. 1024, windows up to the transform
    (2, 60, 2048, 128, {"dft": 4096}),
    (8, 20, 2048, 512, {"dft": 4096}),
    (3, 9, 2048, 256, {"dft": 4096, "chain_len": 8}),
    (2, 9, 1024, 1024, {"dft": 4096, "chain_len": 1}),
    (2, 9, 2048, 1024, {"dft": 4096, "chain_len": 2}),
    (5, 13, 4096, 1024, {"dft": 4096}),
    (2, 7, 4096, 128, {"dft": 4096}),
    (3, 5, 3072, 512, {"dft": 4096, "chain_len": 6}),
    (2, 5, 1536, 512, {"dft": 4096, "chain_len": 2}),
    # the dft 8192 / 16384 team kernels (pv_kernels_team.h): a word per wavefront of a chain, four or eight of them
    (8, 20, 8192, 2048, {"dft": 8192}),
    (2, 30, 4096, 512, {"dft": 8192, "chain_len": 8}),
    (2, 20, 2048, 512, {"dft": 8192, "chain_len": 3}),
    (3, 11, 8192, 4096, {"dft": 8192, "chain_len": 1}),
    (3, 30, 16384, 4096, {"dft": 16384}),
    (2, 30, 4096, 1024, {"dft": 16384, "chain_len": 4}),
    (2, 20, 8192, 1024, {"dft": 16384}),
    # ... at half a step (hop = 64 R samples: half steps leave by the lower 32 lanes) and at hop = window / 16 of a full window
    (2, 30, 4096, 256, {"dft": 8192}),
    (3, 9, 2048, 256, {"dft": 8192, "chain_len": 9}),
    (2, 30, 8192, 512, {"dft": 16384}),
    (2, 9, 4096, 512, {"dft": 16384, "chain_len": 8}),
    (2, 30, 8192, 512, {"dft": 8192}),
    (2, 30, 16384, 1024, {"dft": 16384, "chain_len": 16}),
    # ... at a quarter / an eighth of a step
    (2, 20, 2048, 128, {"dft": 8192}),
    (2, 20, 2048, 128, {"dft": 16384}),
    (3, 5, 4096, 128, {"dft": 8192, "chain_len": 33}),
    (2, 9, 1024, 128, {"dft": 8192, "chain_len": 7}),
])
def test_overlap_fixup_inside_the_kernel_equals_the_separate_launch(fa, ch, seconds, W, hop, hooks):
    """k_synthesize_v2 adds the overlaps of neighbouring chains itself (a tagged word per boundary; the head's owner publishes from inside its frame
    loop, the tail's owner picks the head up under its last transform; whoever finds the other's tag adds -- agent-scope side buffers:
    pv_kernels_v2.h) where its chains are long enough (the library's choice, hook 0), always (hook 1) or never (hook 2: k_ola_fixup in a launch of
    its own).  One addition per sample either way: the outputs must be the same bits, launch after launch."""
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    lib, vp = fa.lib, ctypes.c_void_p
    hooks = dict(hooks)
    sr, dft = 48000.0, hooks.pop("dft", 2048)
    n = int(seconds * sr) + 123
    F = int(lib.flanhip_num_pv_frames(n, hop))
    ar = np.float32(sr) / np.float32(hop)
    bins = dft // 2 + 1
    x = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_noise_dev(vp(x.data_ptr()), ch, n, 77, None))
    pv = torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev)
    fa.check(lib.flanhip_analyze_dev(vp(x.data_ptr()), ch, n, sr, W, hop, dft, vp(pv.data_ptr()), None))
    out_a = torch.empty((ch, F * hop), dtype=torch.float32, device=dev)
    out_b = torch.empty_like(out_a)
    with fa.debug_options(**hooks):
        ws = torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, sr, ar, W), dtype=torch.uint8, device=dev)
        with fa.debug_options(inline_fixup=2):
            fa.synthesize_dev(pv, ch, F, bins, sr, ar, W, out_b, ws, None)
        torch.cuda.synchronize()
        for rep in range(8):
            out_a.fill_(float("nan"))
            with fa.debug_options(inline_fixup=1 if rep < 6 else 0):
                fa.synthesize_dev(pv, ch, F, bins, sr, ar, W, out_a, ws, None)
            torch.cuda.synchronize()
            same = torch.equal(out_a.view(torch.int32), out_b.view(torch.int32))
            if not same:
                bad = (out_a.view(torch.int32) != out_b.view(torch.int32)).nonzero()
                raise AssertionError("launch %d: %d samples differ, first at %s" % (rep, bad.shape[0], bad[0].tolist()))




@pytest.mark.parametrize("ch,seconds,W,hop,dft", [(2, 20.0, 2048, 512, 2048), (3, 7.3, 2048, 512, 2048), (5, 11.1, 1024, 512, 1024), (4, 20.0, 2048, 1024, 2048),
                                                    (2, 30.0, 2048, 128, 4096), (3, 9.0, 4096, 1024, 4096), (2, 30.0, 8192, 2048, 8192), (2, 30.0, 4096, 1024, 16384)])
def test_overlap_protocol_soak_on_two_streams(fa, ch, seconds, W, hop, dft):
    """The chains' overlaps added inside the synthesis kernels (pv_kernels_v2.h / _v3.h: a tagged word per boundary, the head's owner publishing from inside
    its frame loop behind an explicit drain, the tail's owner adding or depositing) rest on in-order retirement and agent-scope stores across the XCDs' private
    L2s.  The long soak lives in tools/soak_fixup.py (30 000 launches, profiles/r05_soak_fixup.json); this is its short form in the suite: 250 fused round
    trips per shape on each of TWO streams at once (a workspace each; launches overlap on the device), every eighth pair of outputs compared BIT FOR BIT with
    the separate-launch form (k_ola_fixup4).  1000 launches per stream over the four shapes."""
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    sr = 48000.0
    n = int(seconds * sr)
    bins = dft // 2 + 1
    F = int(fa.lib.flanhip_num_pv_frames(n, hop))
    ar = sr / hop
    audio = torch.empty((ch, n), dtype=torch.float32, device=dev)
    fa.check(fa.lib.flanhip_noise_dev(ctypes.c_void_p(audio.data_ptr()), ch, n, 4321, None))
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bufs = [(torch.empty((ch, F, bins, 2), dtype=torch.float32, device=dev), torch.empty((ch, F * hop), dtype=torch.float32, device=dev),
             torch.empty(fa.synthesize_workspace_bytes(ch, F, bins, sr, ar, W), dtype=torch.uint8, device=dev)) for _ in streams]

    def step(i):
        pv, out, ws = bufs[i]
        st = int(streams[i].cuda_stream)
        fa.analyze_dev_fused(audio, ch, n, sr, W, hop, dft, pv, ws, st)
        fa.synthesize_dev_fused(pv, ch, F, bins, sr, ar, W, out, ws, None, st)
    launches, bad, checked = 250, 0, 0
    try:
        with fa.debug_options(inline_fixup=2):                       # the separate launch: the yardstick
            step(0)
            torch.cuda.synchronize()
            want = bufs[0][1].clone()
        with fa.debug_options(inline_fixup=1):                       # inside the kernel, whatever the chain length
            for r in range(launches):
                step(0)
                step(1)
                if r % 8 == 7 or r == launches - 1:
                    torch.cuda.synchronize()
                    for i in range(2):
                        checked += 1
                        bad += 0 if torch.equal(bufs[i][1].view(torch.int32), want.view(torch.int32)) else 1
    finally:
        torch.cuda.synchronize()
    print("\n[soak %d ch x %g s (%d, %d, %d)] %d launches on two streams, %d outputs compared, %d differing" % (ch, seconds, W, hop, dft, 2 * launches, checked, bad))
    assert checked >= 60 and bad == 0
