/* flanhip.h -- C ABI of the MI355X (gfx950) phase-vocoder hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  The reference (loganmcbroom/Flan)
 * has no FFI layer -- the path is a set of C++ member functions -- so each entry point below names the reference
 * member function it stands in for (paths relative to /root/reference/src/flan).  The C++ host classes in
 * include/flan/ (flan::Audio, flan::PV, flan::Function) are written on top of exactly these calls; INTEGRATION.md
 * shows the binding a Flan maintainer would add.
 *
 * Conventions
 *   - every function returns FLANHIP_OK (0) or a negative FLANHIP_ERR_*; nothing throws; flanhip_last_error() gives
 *     the text of the last failure on the calling thread.
 *   - host entry points (no suffix) take HOST pointers, do H2D / kernels / D2H on the current device and return
 *     when the result is in the caller's buffer.  The caller owns all host memory.
 *   - device entry points (_dev) take DEVICE pointers plus a hipStream_t passed as void* (NULL = default stream),
 *     enqueue work and return without synchronising; results chain on-device so a
 *     convert_to_PV -> stretch -> convert_to_audio pipeline never leaves HBM.
 *   - layouts are the reference's: audio float[channel][frame] (Audio/AudioBuffer.cpp:479-482),
 *     PV flanhip_MF[channel][frame][bin] (PV/PVBuffer.cpp:526-529).  All indices are 64-bit here (the reference's
 *     int32 product overflows above 2^31 MFs, PV/PVBuffer.cpp:19-22).
 *   - `cancel` (may be NULL) is the reference's canceller (defines.h:49-62): polled between kernel batches; when it
 *     reads non-zero the call stops and returns FLANHIP_ERR_CANCELLED (the C++ layer then returns a null object).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     FLANHIP_ERR_NO_DEVICE.
 */
#ifndef FLANHIP_H
#define FLANHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLANHIP_OK                 0
#define FLANHIP_ERR_INVALID_ARG   -1   /* null pointer, non-positive size, window > dft ...                    */
#define FLANHIP_ERR_UNSUPPORTED   -2   /* dft size odd, below 4 or above 2^20, chain constraints ...           */
#define FLANHIP_ERR_HIP           -3   /* a HIP runtime call failed; see flanhip_last_error()                  */
#define FLANHIP_ERR_CANCELLED     -4   /* the canceller was raised                                              */
#define FLANHIP_ERR_NO_DEVICE     -5   /* no gfx950 device visible                                              */

/* defines.h:31-35  struct MF { Magnitude m; Frequency f; } */
typedef struct flanhip_MF { float m; float f; } flanhip_MF;

/* ---- library / device ------------------------------------------------------------------------------------ */
int          flanhip_version(void);                 /* 10000*major + 100*minor + patch */
const char * flanhip_last_error(void);
int          flanhip_device_count(void);            /* 0 when no device; never fails   */
int          flanhip_set_device(int device);
int          flanhip_get_device(int * device);      /* the calling thread's current device */

/* ---- shape helpers (pure host arithmetic, usable without a device) ---------------------------------------- */
/* Conversions/AudioPV.cpp:17   numHops = ceil( num_frames / hop ) + 1, INTEGER division */
int64_t flanhip_num_pv_frames(int64_t num_audio_frames, int hop);
/* PV/PVBuffer.cpp:381-384      get_hop_size() = Frame( sample_rate / analysis_rate )     */
int     flanhip_hop_size(float sample_rate, float analysis_rate);
/* PV/PVModify.cpp:312-316      ceil( time_to_frame( max of the time map ) ), from a host grid float[F][bins] */
int64_t flanhip_modify_time_out_frames(const float * mod_seconds, int64_t num_frames, int num_bins, float sample_rate, int hop);

/* ---- device memory plumbing (so that C / C++ / ctypes callers need no HIP headers) ------------------------ */
int flanhip_malloc(void ** dptr, size_t bytes);
int flanhip_free(void * dptr);
int flanhip_memcpy_h2d(void * dst, const void * src, size_t bytes, void * stream);
int flanhip_memcpy_d2h(void * dst, const void * src, size_t bytes, void * stream);
int flanhip_memset(void * dst, int value, size_t bytes, void * stream);
int flanhip_stream_synchronize(void * stream);
/* Cancellation INSIDE a launch (defines.h:49-62: the reference polls its flag once per frame, AudioPV.cpp:49,115).  Wait for `stream` like
 * flanhip_stream_synchronize, but poll the caller's flag meanwhile; when it rises, the conversion kernels this thread has launched stop
 * starting chains -- a block of the FFT kernels walks at most 512 frames (~3 ms), the direct-sum kernels look every few batches of frames --
 * and the call returns FLANHIP_ERR_CANCELLED once the stream has drained (outputs are then unspecified).  Returns FLANHIP_OK when the work
 * completed without the flag rising.  The _fn form takes a predicate (non-zero = cancel) instead of an int flag, for callers whose flag is
 * not an int (the C++ classes' std::atomic<bool>&).  The host-buffer entry points (flanhip_analyze, flanhip_synthesize) wait this way with
 * their own `cancel` argument. */
/* Scope: the wait on `stream` stops the conversion kernels THIS thread has launched on THAT stream; what the thread has in flight on its
 * other streams (and what other threads have in flight anywhere) runs to completion.  (A thread that uses more than 64 streams at once on
 * one device shares cancel words between them: such a stream may be stopped together with another one of the same thread.) */
int flanhip_wait_cancellable(void * stream, volatile int * cancel);
int flanhip_wait_cancellable_fn(void * stream, int (*poll)(void * user), void * user);
/* Transfers between ordinary (pageable) host memory and the device, synchronous: what the host entry points below
 * (flanhip_analyze, flanhip_synthesize, ...) use for the caller's buffers.  A download first lets a pool of worker threads fault
 * the destination's pages in together (a fresh allocation costs more to fault in on one thread than to fill over the link), then
 * both directions are the runtime's copies, which run at the link's rate.  flanhip_touch_pages is that first step on its own
 * (it writes a zero into every page: for memory about to be overwritten).  flanhip_parallel_for runs fn( ctx, i ), i in [0, n), on
 * those workers and the calling thread (as many threads as the process may use; FLAN_HOST_THREADS overrides); a region started
 * while another runs, or from inside one, runs inline.  fn must not throw. */
int flanhip_upload(void * d_dst, const void * src, size_t bytes);
int flanhip_download(void * dst, const void * d_src, size_t bytes);
int flanhip_touch_pages(void * ptr, size_t bytes);
int flanhip_host_workers(void);
int flanhip_parallel_for(int n_tasks, void (*fn)(void *, int), void * ctx);
/* a stream of the caller's own (e.g. one per copy direction, so that a download and an upload overlap); every entry point
 * that takes `void * stream` accepts it, NULL stays the default stream */
int flanhip_stream_create(void ** stream);
int flanhip_stream_destroy(void * stream);
/* page-locked host memory: what a caller samples a Function grid into (Function.h:155-171) so that the upload runs at the
 * link's rate instead of through the runtime's staging copy.  Needs a device. */
int flanhip_host_malloc(void ** hptr, size_t bytes);
int flanhip_host_free(void * hptr);

/* ---- Audio::convert_to_PV  (Conversions/AudioPV.cpp:12-78, phase_vocoder.cpp:5-53, WindowFunctions.cpp:10-13) - */
/* audio: float[ch][n]; out: MF[ch][F][dft/2+1] with F = flanhip_num_pv_frames(n, hop), written to *num_pv_frames. */
/* dft_size: any even size in [4, 2^20] with window_size <= dft_size (FFTHelper.cpp:16-26 hands the caller's size to FFTW as it is).  Which
 * kernels serve a size: powers of two in [32, 16384] tuned or LDS-resident FFT kernels; other sizes whose half factors
 * into 2 ... 13, up to 16384, mixed-radix FFT kernels; sizes above 16384 whose half is C1 <= 256 times a product of 2 ... 13 up to 4096 (32768 ... 2^20,
 * 20000, 44100, 48000 ...) residue-pair kernels; every other size whose half is 64 ... 131072 (a larger prime factor) Bluestein's chirp-z form -- in LDS up
 * to dft 8192, in device memory above (convert_to_PV takes that stretch from the stream: hipMallocAsync / hipFreeAsync around the launch); anything else
 * the transform's definition summed in fp64 (pv_kernels_any.h): O( window x bins ) per frame, same results.  flanhip_synthesize_workspace_bytes includes
 * the scratch a size needs (direct sums: one PV's worth of spectra + frames x window floats; residue pairs: C1 / 2 + 1 output streams; chirp-z in
 * device memory: two frames of the transform's length per chain). */
int flanhip_analyze(const float * audio, int64_t num_channels, int64_t num_audio_frames, float sample_rate,
                    int window_size, int hop, int dft_size,
                    flanhip_MF * out, int64_t * num_pv_frames, volatile int * cancel);
int flanhip_analyze_dev(const float * d_audio, int64_t num_channels, int64_t num_audio_frames, float sample_rate,
                        int window_size, int hop, int dft_size,
                        flanhip_MF * d_out, void * stream);

/* ---- PV::convert_to_audio  (Conversions/AudioPV.cpp:86-139, phase_vocoder.cpp:55-61) ------------------------- */
/* pv: MF[ch][F][bins]; out: float[ch][F*hop], hop = flanhip_hop_size(sr, analysis_rate).
 * *nan_flag (may be NULL) is set to 1 when the buffer holds a NaN/Inf (PV/PVBuffer.cpp:44-50; the reference prints
 * a warning and carries on, AudioPV.cpp:88-89 -- so do we). */
int flanhip_synthesize(const flanhip_MF * pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                       float sample_rate, float analysis_rate, int window_size,
                       float * out, int * nan_flag, volatile int * cancel);
/* Device form.  d_workspace must hold flanhip_synthesize_workspace_bytes(...) bytes for the SAME arguments (and the same
 * setting of the debug knobs FLANHIP_CHAIN_LEN / FLANHIP_TARGET_CHAINS / FLANHIP_FORCE_GENERIC, which change the layout and
 * are read per call); d_nan_flag (may be NULL) is a device int that is OR-ed with 1 on NaN/Inf (the caller zeroes it), and with 2 by
 * flanhip_synthesize_dev_fused when the workspace does not hold what the library noted its last producer left there (two callers racing on one
 * workspace between an analysis and its synthesis: the output is then not to be trusted). */
size_t flanhip_synthesize_workspace_bytes(int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                          float sample_rate, float analysis_rate, int window_size);
int flanhip_synthesize_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                           float sample_rate, float analysis_rate, int window_size,
                           float * d_out, void * d_workspace, int * d_nan_flag, void * stream);

/* Fused round trip (convert_to_PV immediately followed, possibly after read-only use, by convert_to_audio of the SAME,
 * unmodified PV): the analysis kernel also leaves in `d_synth_workspace` (sized by flanhip_synthesize_workspace_bytes for the
 * PV it produces) what synthesis' pre-pass would compute -- the per-chain sums of the phase increments and the NaN/Inf flag --
 * and flanhip_synthesize_dev_fused starts from them instead of re-reading the PV.  Results are identical to the unfused pair. */
int flanhip_analyze_dev_fused(const float * d_audio, int64_t num_channels, int64_t num_audio_frames, float sample_rate,
                              int window_size, int hop, int dft_size,
                              flanhip_MF * d_out, void * d_synth_workspace, void * stream);
int flanhip_synthesize_dev_fused(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                 float sample_rate, float analysis_rate, int window_size,
                                 float * d_out, void * d_workspace, int * d_nan_flag, void * stream);
/* ... and for a PV whose workspace MAY hold the pre-pass (flanhip_modify_time_dev_fused): the pre-pass kernel is launched and
 * retires at once when the sums are there.  The hand-over is consumed: a second call on the same workspace runs the pre-pass. */
int flanhip_synthesize_dev_fused_checked(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                         float sample_rate, float analysis_rate, int window_size,
                                         float * d_out, void * d_workspace, int * d_nan_flag, void * stream);

/* The same call with the kernels it launches as a per-call argument: presummed 0 = flanhip_synthesize_dev, 1 = _fused, 2 = _fused_checked;
 * stages: bit 0 k_phase_sums, 1 k_phase_scan, 2 k_synthesize, 3 k_ola_fixup (0xF: all of them, i.e. the plain entry points above).
 * For timing ONE kernel of a call with events (bench.py's roofline object); the output is only meaningful with all four. */
int flanhip_synthesize_dev_stages(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                  float sample_rate, float analysis_rate, int window_size,
                                  float * d_out, void * d_workspace, int * d_nan_flag, int presummed, int stages, void * stream);

/* Test / A-B hooks, PER CALLING THREAD (thread-local, all 0 = off by default; the library's own paths and the C++ classes never set them, and
 * no entry point reads an environment variable).  A thread that sets one changes how ITS later calls are cut or routed -- never the results
 * beyond the documented tolerances (tests/test_gpu_conversions.py holds that) -- and must size a workspace under the same settings it
 * converts with (chain length and chain count decide the layout). */
#define FLANHIP_DEBUG_CHAIN_LEN       0   /* frames per chain */
#define FLANHIP_DEBUG_TARGET_CHAINS   1   /* chains the tuned kernels are cut for (default: what the device holds at once) */
#define FLANHIP_DEBUG_FORCE_GENERIC   2   /* 1: never the tuned dft 2048 / 4096 kernels */
#define FLANHIP_DEBUG_NO_FAST_DIV     3   /* 1: hardware division by the analysis rate */
#define FLANHIP_DEBUG_ANA_VARIANT     4   /* dft 2048 analysis: ablated instantiations (diagnostic builds only) */
#define FLANHIP_DEBUG_SYN_VARIANT     5   /* dft 2048 synthesis: 2 = behind the scan kernel even where it could work out its own carries; ablations */
#define FLANHIP_DEBUG_ANA4096_OLD     6   /* 1: the generic block-per-chain analysis kernel (pv_kernels.h) instead of the dft 4096 team kernel (A/B predecessor) */
#define FLANHIP_DEBUG_SYN4096_OLD     7   /* 1: the generic dft 4096 synthesis kernel */
#define FLANHIP_DEBUG_RESAMPLE_DIRECT 8   /* 1: Audio::resample's 2:1 block convolver always as direct fp64 sums in the checker's operation order
                                           * (default: fp64 overlap-save FFT convolution, the reference's own method, r8brain/CDSPBlockConvolver.h:242-344,
                                           * for float streams of at least 8 blocks); 2: the convolver's 256-thread radix-16 generation (A/B predecessor) */
#define FLANHIP_DEBUG_FORCE_DIRECT    9   /* 1: dft sizes without power-of-two kernels as direct fp64 sums (the transform's definition: pv_kernels_any.h),
                                           * never the mixed-radix FFT kernels (pv_kernels_mr.h): the checker-order path, for A/B */
#define FLANHIP_DEBUG_INLINE_FIXUP   10   /* the dft 512 ... 16384 synthesis kernels adding the overlaps of neighbouring chains themselves (a tagged word per boundary, the
                                           * head's owner publishing from inside its frame loop; agent-scope side buffers) instead of k_ola_fixup in a launch of
                                           * its own -- the same sums.  0: where the chains are long enough (the default), 1: always, 2: never */
#define FLANHIP_DEBUG_WIDE_OFFSETS   11   /* 1: kernels that choose 32-bit element offsets for grids below 2^30 elements (k_stretch_map) take their 64-bit
                                           * form whatever the size: the path of multi-gigabyte grids, testable on small ones */
#define FLANHIP_DEBUG_NO_SUB         12   /* 1: dft 512 / 256 / 128 never on the kernels with several chains per wavefront (pv_kernels_sub.h, round 6): A/B against
                                             the one-wavefront kernels (dft 512) / the generic ones (dft 256) */
void flanhip_debug_option(int which, int value);
/* Scratch (private memory) bytes per lane of a kernel whose hand-counted s_waitcnt values are only right while the compiler emits no memory
 * operation of its own on that path -- a spill or reload is one (tests/test_gpu_processors.py holds them to 0).  which: 0 / 1 = k_stretch_map with
 * 32-bit / 64-bit row offsets.  Negative: error. */
int flanhip_debug_kernel_scratch_bytes(int which);

/* ---- PV frame processors ------------------------------------------------------------------------------------- */
/* modify_time_base (PV/PVModify.cpp:307-362, linear Interpolator): mod_seconds is the sampled time map float[F][bins]
 * (FunctionSample2d layout, FunctionSample.h:173-199).  out: MF[ch][out_frames][bins], out_frames from
 * flanhip_modify_time_out_frames(). */
int flanhip_modify_time(const flanhip_MF * pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                        float sample_rate, int hop, const float * mod_seconds,
                        int64_t out_frames, flanhip_MF * out, volatile int * cancel);
int flanhip_modify_time_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                            float sample_rate, int hop, const float * d_mod_seconds,
                            int64_t out_frames, flanhip_MF * d_out, void * stream);
/* PV::stretch front half (PV/PVModify.cpp:371-382): in-place inclusive prefix sum over frames per bin (fp32, sequential
 * order) then frame_to_time.  d_factor: float[F][bins] factor grid in, seconds out; flanhip_stretch_map_dev (below) also
 * reduces the maximum of the result into *d_max (float) when that pointer is given. */
/* modify_time for a PV that goes on to convert_to_audio: d_workspace is a synthesis workspace for the OUTPUT PV
 * (flanhip_synthesize_workspace_bytes( ch, out_frames, bins, sr, analysis_rate, window_size ); hop = int( sr / analysis_rate )).  When the time map never runs
 * backwards (every stretch) the kernel that writes the output also leaves convert_to_audio's pre-pass there;
 * flanhip_synthesize_dev_fused_checked then skips its own.  Same output as flanhip_modify_time_dev in every case. */
int flanhip_modify_time_dev_fused(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                  float sample_rate, float analysis_rate, const float * d_mod, int64_t out_frames,
                                  flanhip_MF * d_out, int window_size, void * d_workspace, void * stream);
int flanhip_stretch_map_dev(float * d_factor, int64_t num_pv_frames, int num_bins, float sample_rate, int hop,
                            float * d_max, void * stream);
/* PV::stretch with a CONSTANT factor: the whole map (running sum down the frames + frame_to_time, and its maximum) from the factor alone -- the sum
 * of a constant has a closed form that reproduces the sequential fp32 additions bit for bit (flan_amd/csrc/const_sum.h), so nothing is filled and
 * nothing is scanned.  d_map: float[F][bins] out.  Same result as flanhip_fill_dev + flanhip_stretch_map_dev. */
int flanhip_stretch_map_const_dev(float factor, float * d_map, int64_t num_pv_frames, int num_bins, float sample_rate, int hop,
                                  float * d_max, void * stream);
/* A grid filled with one value on the device (a constant-valued Function, sampled: `[](TF){ return c; }`). */
int flanhip_fill_dev(float * d_grid, int64_t count, float value, void * stream);

/* modify_frequency_base (PV/PVModify.cpp:196-257): mod_hz float[F][bins] = where each grid bin centre maps;
 * in_modified float[ch][F][bins] = new frequency of every MF.  out: MF[ch][F][bins]. */
int flanhip_modify_frequency(const flanhip_MF * pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                             float sample_rate, const float * mod_hz, const float * in_modified,
                             flanhip_MF * out, volatile int * cancel);
int flanhip_modify_frequency_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                 float sample_rate, const float * d_mod_hz, const float * d_in_modified,
                                 flanhip_MF * d_out, void * stream);
/* PV::repitch front half (PV/PVModify.cpp:273-302): d_factor float[F][bins] factor grid in -> Hz map out;
 * d_in_modified float[ch][F][bins] out. */
int flanhip_repitch_map_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                            float sample_rate, float * d_factor, float * d_in_modified, void * stream);
/* PV::repitch in one piece (PV/PVModify.cpp:273-305): d_factor float[F][bins] is turned into the Hz map in place (the running sum
 * over bins, :278-284) and modify_frequency_base runs with every MF's target frequency looked up on the fly (:289-302) -- the
 * same results as flanhip_repitch_map_dev + flanhip_modify_frequency_dev without the intermediate float[ch][F][bins] grid. */
int flanhip_repitch_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins, float sample_rate,
                        float * d_factor, flanhip_MF * d_out, void * stream);
/* The same four with a named non-linear Interpolator (PVModify.cpp:232 and :344 apply interp( ... ) to the mixing coordinate of every
 * output bin / frame; Utility/Interpolator.cpp:14-101).  interp: FLANHIP_INTERP_* (below); FLANHIP_INTERP_LINEAR gives the plain calls
 * bit for bit. */
int flanhip_modify_time_interp_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                   float sample_rate, int hop, const float * d_mod_seconds,
                                   int64_t out_frames, int interp, flanhip_MF * d_out, void * stream);
int flanhip_modify_time_interp_dev_fused(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                         float sample_rate, float analysis_rate, const float * d_mod_seconds,
                                         int64_t out_frames, int interp, flanhip_MF * d_out, int window_size, void * d_synth_workspace, void * stream);
int flanhip_modify_frequency_interp_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                        float sample_rate, const float * d_mod_hz, const float * d_in_modified,
                                        int interp, flanhip_MF * d_out, void * stream);
int flanhip_repitch_interp_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins, float sample_rate,
                               float * d_factor, int interp, flanhip_MF * d_out, void * stream);


/* PV::shape (PV/PV.cpp:421-458) for the affine shaper  mf -> { a*m + b, c*f + d }; arbitrary host callables are
 * evaluated by the C++ layer on the host grid and uploaded through flanhip_shape_table_dev. */
int flanhip_shape_affine(const flanhip_MF * pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                         float sample_rate, float a, float b, float c, float d, int use_shift_alignment,
                         flanhip_MF * out, volatile int * cancel);
int flanhip_shape_affine_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                             float sample_rate, float a, float b, float c, float d, int use_shift_alignment,
                             flanhip_MF * d_out, void * stream);
/* Same placement rule with the shaped values precomputed (d_shaped: MF[ch][F][bins] = shaper(in) per MF). */
int flanhip_shape_table_dev(const flanhip_MF * d_pv, const flanhip_MF * d_shaped, int64_t num_channels,
                            int64_t num_pv_frames, int num_bins, float sample_rate, int use_shift_alignment,
                            flanhip_MF * d_out, void * stream);
/* The same without shift alignment, leaving PV::convert_to_audio's pre-pass for the RESULT in a synthesis workspace (d_ws of
 * flanhip_synthesize_workspace_bytes for the result's shape): follow with flanhip_synthesize_dev_fused on d_out. */
int flanhip_shape_affine_dev_fused(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float sample_rate,
                                   float analysis_rate, float a, float b, float c, float d, flanhip_MF * d_out, int window_size,
                                   void * d_ws, void * stream);
int flanhip_shape_table_dev_fused(const flanhip_MF * d_pv, const flanhip_MF * d_shaped, int64_t num_channels, int64_t num_frames,
                                  int num_bins, float sample_rate, float analysis_rate, flanhip_MF * d_out, int window_size,
                                  void * d_ws, void * stream);

/* ---- further PV frame processors.  User functions are SAMPLED BY THE CALLER, as the reference samples
 * them on the host before its loops (PV.h:31-35, Function.h:141-171): a grid pointer, or NULL plus a constant. ------------- */

/* Utility/Interpolator.cpp:14-101: the named interpolators a device kernel can evaluate itself */
#define FLANHIP_INTERP_LINEAR        0
#define FLANHIP_INTERP_MIDPOINT      1
#define FLANHIP_INTERP_NEAREST       2
#define FLANHIP_INTERP_FLOOR         3
#define FLANHIP_INTERP_CEIL          4
#define FLANHIP_INTERP_SMOOTHSTEP    5
#define FLANHIP_INTERP_SMOOTHERSTEP  6
#define FLANHIP_INTERP_SQRT          7
#define FLANHIP_INTERP_SINE          8   /* cosf based: agrees with the host libm to 1-2 ulp, not bit for bit */
/* An Interpolator built from an arbitrary callable (Utility/Interpolator.h: Interpolator( std::function<float(float)> )): the caller samples it
 * at i / FLANHIP_INTERP_TABLE_INTERVALS, i = 0 .. INTERVALS, plus its value at NaN (INTERVALS + 2 floats, host memory), and gets a kind
 * (>= FLANHIP_INTERP_TABLE_FIRST, at most 32 alive per process) that every entry point taking FLANHIP_INTERP_* accepts ON THE DEVICE THAT WAS
 * CURRENT AT CREATION -- on another device the kind is rejected as unknown (create one table per device).  The kernels read the
 * table with linear interpolation between neighbouring samples, argument clamped to [0, 1]: exact at the sample points, within
 * max|f''| / ( 8 INTERVALS^2 ) = 2.9e-11 max|f''| between them -- below one fp32 step for any smooth shaping curve; a jump of the callable is
 * smeared over one interval (1.5e-5 wide).  _destroy synchronises the table's own device before the table is freed, whichever device is current. */
#define FLANHIP_INTERP_TABLE_INTERVALS 65536
#define FLANHIP_INTERP_TABLE_FIRST     16
int flanhip_interp_table_create(const float * samples, int * kind);
int flanhip_interp_table_destroy(int kind);

/* PV::replace_amplitudes (PV/PV.cpp:205-236): out = { src.m * a + m * (1-a), f }, a = clamp(amount,0,1), on the overlap of the
 * two PVs; zero elsewhere.  d_amount: float[F][bins] over THIS pv's domain, or NULL to use amount_const. */
int flanhip_replace_amplitudes_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins,
                                   const flanhip_MF * d_src, int64_t src_channels, int64_t src_frames, int src_bins,
                                   const float * d_amount, float amount_const, flanhip_MF * d_out, void * stream);
/* PV::subtract_amplitudes (PV/PV.cpp:238-264): out = copy, m = |m - src.m * amount| on the overlap (amount not clamped) */
int flanhip_subtract_amplitudes_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins,
                                    const flanhip_MF * d_src, int64_t src_channels, int64_t src_frames, int src_bins,
                                    const float * d_amount, float amount_const, flanhip_MF * d_out, void * stream);

/* PV::resonate (PV/PV.cpp:604-641).  Output frames (:613) = num_frames + ceil( time_to_frame( max(length,0) ) ). */
int64_t flanhip_resonate_out_frames(int64_t num_frames, float length_seconds, float sample_rate, int hop);
/* d_decay: float[out_frames][bins] sampled over the OUTPUT's domain (:616), or NULL to use decay_const; clamped to [0,1] (:617).
 * The per-frame factor pow(decay, seconds per frame) (:631) is the host libm's powf for a constant (bit for bit the
 * reference's call on this platform) and the correctly rounded power for a grid. */
int flanhip_resonate_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float sample_rate, int hop,
                         int64_t out_frames, const float * d_decay, float decay_const, flanhip_MF * d_out, void * stream);

/* PV::retain_n_loudest_partials / remove_n_loudest_partials (PV/PV.cpp:552-602).  d_n: int32[num_frames], the sampled
 * Function<Second,Bin> (:555), or NULL to use n_const; clamped to [0, num_frames] as the reference does (:556).
 * Bins of equal |m| rank by ascending bin (the reference's std::sort leaves their order unspecified). */
int flanhip_n_loudest_partials_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins,
                                   const int32_t * d_n, int32_t n_const, int remove, flanhip_MF * d_out, void * stream);

/* PV::desample (PV/PVModify.cpp:445-511).  d_ratio: float[F][bins] or NULL + ratio_const; interp: FLANHIP_INTERP_* */
int flanhip_desample_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins,
                         const float * d_ratio, float ratio_const, int interp, flanhip_MF * d_out, void * stream);

/* PV::time_extrapolate (PV/PVModify.cpp:607-666) after its input validation: 0 <= start_frame < end_frame < num_frames,
 * out_frames = end_frame + extrapolated frames (:627), d_interp_samples: float[out_frames - start_frame] sampled by the caller
 * as :631-633 does.  d_out: MF[ch][out_frames][bins]. */
int flanhip_time_extrapolate_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float sample_rate,
                                 int64_t start_frame, int64_t end_frame, int64_t out_frames, const float * d_interp_samples,
                                 flanhip_MF * d_out, void * stream);

/* ---- PV methods that select, rearrange and re-place frames and bins (PV/PV.cpp:24-39, :92-198, :362-419, :643-720) ---- */
/* PV::get_frame (PV/PV.cpp:24-39): one frame interpolated between its neighbours (getBinInterpolated, :62-73).
 * frame_pos = clamp( time_to_frame( time ), 0, F-1 ) is the caller's (:28); d_out: MF[ch][1][bins]. */
int flanhip_get_frame_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float frame_pos,
                          int interp, flanhip_MF * d_out, void * stream);
/* out[c][o][:] = pv[c][src[o]][:], zero where src[o] < 0: the copy loops of PV::freeze (PV/PV.cpp:176-195).
 * d_src_frames: int32[out_frames] (flanhip_freeze_plan fills the host copy). */
int flanhip_select_frames_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins,
                              const int32_t * d_src_frames, int64_t out_frames, flanhip_MF * d_out, void * stream);
/* PV::freeze's timing logic (PV/PV.cpp:140-171), host arithmetic: the output's frame count (returned; -1 on bad arguments) and,
 * when src_frames is not NULL, the input frame every output frame repeats (-1: stays zero).  times / lengths: n pairs, seconds.
 * Of several events on one frame the first given survives (unspecified in the reference: its sort is not stable). */
int64_t flanhip_freeze_plan(int64_t num_frames, float sample_rate, int hop, const float * times, const float * lengths, int n,
                            int32_t * src_frames);
/* PV::cut_frames (PV/PV.cpp:643-668): its input validation (count 0 = the null PV it returns), and the copy. */
int flanhip_cut_frames_range(int64_t num_frames, int32_t start, int32_t end, int32_t * start_out, int32_t * count_out);
int flanhip_cut_frames_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, int64_t start,
                           int64_t count, flanhip_MF * d_out, void * stream);
/* one input of PV::join (PV/PV.cpp:708-716): its frames to out frames [out_start, out_start + in_frames), the channels and bins
 * both have; the caller clears d_out first (:706) */
int flanhip_place_frames_dev(const flanhip_MF * d_in, int64_t in_channels, int64_t in_frames, int in_bins, flanhip_MF * d_out,
                             int64_t out_channels, int64_t out_frames, int out_bins, int64_t out_start, void * stream);
/* PV::select (PV/PV.cpp:92-127).  d_selector_tf: TF{ t, f }[out_frames][bins], the selector sampled over the OUTPUT's domain
 * (:103); out_frames = Frame( time_to_frame( length ) ) (:101).  d_out: MF[ch][out_frames][bins]. */
int flanhip_select_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float sample_rate, int hop,
                       const float * d_selector_tf, int64_t out_frames, flanhip_MF * d_out, void * stream);
/* PV::add_octaves (mode 0) / PV::add_harmonics (mode 1) (PV/PV.cpp:362-419).  d_series: float[F][num_harmonics], the series
 * callable at ( frame_to_time( frame ), harmonic ) for the 0-based harmonic index, as :371-379 samples it; num_harmonics =
 * ceil( log2( get_height() ) ) for octaves (:412), num_bins for harmonics (:418).  dft sizes up to 8192. */
int flanhip_harmonic_scale_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float sample_rate,
                               const float * d_series, int num_harmonics, int mode, flanhip_MF * d_out, void * stream);

/* PV::modify (PV/PVModify.cpp:15-193): the general time / frequency warp.  The caller samples the callable twice, as the
 * reference does:  d_mod_tf: TF{ t, f }[F][bins], mod over the input's grid (:22), seconds / Hz;  d_in_f: float[ch][F][bins],
 * mod( { frame_to_time( frame ), the MF's own frequency } ).f (:62-66).  interp: FLANHIP_INTERP_*.
 * flanhip_modify_out_frames (host arithmetic, host grid): the output's frame count (:28-38); -2 when it would be longer than 10
 * minutes (the reference prints a message and returns a null PV, :30-34), 0 when there is nothing to make, -1 on bad arguments.
 * Equally loud candidates for an output point: the first input quad in ( frame, bin ) order gives the frequency (unspecified in the
 * reference, which runs frames in parallel); quads with a non-finite corner offer nothing. */
int64_t flanhip_modify_out_frames(const float * mod_tf, int64_t num_frames, int num_bins, float sample_rate, int hop);
int flanhip_modify_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float sample_rate, int hop,
                       const float * d_mod_tf, const float * d_in_f, int interp, int64_t out_frames, flanhip_MF * d_out, void * stream);

/* PV::stretch_spline (PV/PVModify.cpp:387-443; the spline is the reference's vendored spline/spline.h:284-401, fp64).
 * steps: HOST uint32[F-1], safeInterpolation( frame ) = max( uint32( interpolation( frame * frame_to_time( 1 ) ) ), 1 ) of every
 * frame but the last (:391-394); the output has sum( steps ) frames (:399-405, flanhip_stretch_spline_out_frames; -1 when F < 3,
 * a step is 0 or the sum does not fit a Frame).  Allocates its workspace (8 B x F x ch x bins x 2) from the stream's memory pool. */
int64_t flanhip_stretch_spline_out_frames(const uint32_t * steps, int64_t num_frames);
int flanhip_stretch_spline_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins,
                               const uint32_t * steps, int64_t out_frames, flanhip_MF * d_out, void * stream);

/* PV::smear_time (PV/PVModify.cpp:513-605).  The caller samples the three callables as :520-524 and :558-560 do:
 *   smear: float[F][bins] seconds (NULL: smear_const), clamped to >= 0 inside;  granularity: int32[F][bins] (NULL: the constant),
 *   clamped to >= 1 inside;  distribution: float[2 * dist_samples_2], distribution( x / dist_samples_2 ), x in [-dist_samples_2, dist_samples_2).
 * flanhip_smear_time_plan (host arithmetic, host grid): the frame the output starts at, its frame count (:563) and dist_samples_2 (:555-556). */
int flanhip_smear_time_plan(int64_t num_frames, int num_bins, float sample_rate, int hop, const float * smear, float smear_const,
                            int32_t * true_left, int64_t * out_frames, int32_t * dist_samples_2);
int flanhip_smear_time_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_frames, int num_bins, float sample_rate, int hop,
                           const float * d_smear, float smear_const, const int32_t * d_granularity, int32_t granularity_const,
                           const float * d_distribution, int64_t n_distribution, int32_t true_left, int64_t out_frames,
                           flanhip_MF * d_out, void * stream);

/* ---- Audio::convert_to_mid_side / convert_to_left_right (Audio/AudioConversions.cpp:32-56), stereo only ------ */
int flanhip_mid_side_dev(const float * d_in, int64_t num_audio_frames, float * d_out, void * stream);

/* ---- Audio::resample (Audio/AudioConversions.cpp:14-30, r8brain CDSPResampler with default parameters) --------- */
/* AudioConversions.cpp:22: out frames = Frame( float(num_frames) * ( dst_rate / src_rate ) ) */
int64_t flanhip_resample_out_frames(int64_t num_frames, float src_rate, float dst_rate);
/* in: float[ch][n]; out: float[ch][flanhip_resample_out_frames(n,...)].  Like the reference, the whole channel-major buffer is
 * resampled as ONE stream (filter ringing crosses channel boundaries).  Every chain CDSPResampler builds for two positive rates is served
 * (r8brain/CDSPResampler.h:119-378): one block convolver (2:1 -- 96 -> 48 kHz, the tuned kernel --, 3:1, 3:2, 2:3, 4:3, 1:2, 1:3); block
 * convolver + half-band upsamplers (4x, 8x ..., 6x, 12x ...); 2x convolver + fractional interpolator (44.1 <-> 48 kHz ...: whole stepping, or the
 * spline-interpolated bank for rates without a small common divisor, e.g. 44.1 kHz -> 48001 Hz); intermediate interpolation (8 -> 44.1 kHz,
 * 44.1 -> 192 kHz: 2x convolver, interpolator, a 2x convolver whose transition band follows from the rates, half-band upsamplers); half-band
 * downsamplers + convolver [+ interpolator] (192 -> 48, 96 -> 16, 192 -> 44.1 kHz ...).  Chains of several stages take transient fp64 streams
 * from the stream's memory pool; a chain with the spline bank synchronises the stream once (a small table goes up) and, like the reference,
 * its result depends on num_frames (r8brain re-bases the interpolator's position counter once per process() call = per num_frames input
 * samples).  Results: bit-identical to the vendored r8brain on the committed vectors (tests/golden/ref_made/r8brain.npz) but for samples
 * within rounding of a tie (>= 99.9 % identical, the rest one fp32 ulp). */
int flanhip_resample(const float * in, int64_t num_channels, int64_t num_frames, float src_rate, float dst_rate,
                     float * out, volatile int * cancel);
int flanhip_resample_dev(const float * d_in, int64_t num_channels, int64_t num_frames, float src_rate, float dst_rate,
                         float * d_out, void * stream);

/* ---- a PV that is a FRAME RANGE of a longer one (few channels, long signals: frame ranges instead of channels shard across
 * GPUs).  Analysis needs nothing new (a range of frames depends only on the samples under its windows and on the
 * frame before it).  Synthesis integrates the phase over ALL earlier frames, so it runs in two steps with one small exchange
 * between them:  (1) every rank: flanhip_synthesize_prepass_dev -> d_total_out[ch][bins], the phase its own frames add up to
 * (mod pi2), the chain sums stay in the workspace;  (2) the ranks exchange the totals; rank r folds the totals of ranks < r into
 * d_carry_in and calls flanhip_synthesize_dev_carry.  Samples within window/2 of a range boundary receive contributions from
 * both neighbours: flan_amd/sharding.py pads each range with silent frames so that they land inside the local output, and adds
 * the overlaps.  NaN flag as in flanhip_synthesize_dev (step 1 scans the data). ------------------------------------------------- */
int flanhip_synthesize_prepass_dev(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                   float sample_rate, float analysis_rate, int window_size, void * d_workspace,
                                   double * d_total_out, int * d_nan_flag, void * stream);
int flanhip_synthesize_dev_carry(const flanhip_MF * d_pv, int64_t num_channels, int64_t num_pv_frames, int num_bins,
                                 float sample_rate, float analysis_rate, int window_size, float * d_out, void * d_workspace,
                                 const double * d_carry_in, int * d_nan_flag, void * stream);

/* ---- multi-GPU: reassembling the channel shards of the output.  One process per GPU; channels shard with no
 * exchange during compute; the output step is ONE in-place all-gather over RCCL / xGMI.  RCCL is bound at run time (dlopen):
 * without librccl.so these four return FLANHIP_ERR_UNSUPPORTED and everything else still works. ------------------------- */
#define FLANHIP_COMM_ID_BYTES 128                     /* ncclUniqueId */
/* rank 0 creates the id and hands the 128 bytes to the other ranks by whatever launcher it uses (MPI, a file, a socket ...) */
int flanhip_comm_unique_id(char * id_out /* [FLANHIP_COMM_ID_BYTES] */);
/* collective over all ranks; the device is the one current in this process (flanhip_set_device).  comm_out: an ncclComm_t */
int flanhip_comm_init(const char * id, int world_size, int rank, void ** comm_out);
int flanhip_comm_destroy(void * comm);
/* d_all: float[world_size][count_per_rank]; this rank's shard (its channels, Audio layout float[ch][frames]) is already at
 * d_all + rank * count_per_rank.  On return (stream order) d_all is the whole float[all channels][frames] buffer on every rank.
 * comm may also be an ncclComm_t created elsewhere with the same RCCL. */
int flanhip_allgather_audio(void * comm, float * d_all, int64_t count_per_rank, int rank, void * stream);

/* ---- synthetic input + comparison utilities (bench / tests; defined by this project) -------------- */
int flanhip_noise_dev(float * d_out, int64_t num_channels, int64_t num_audio_frames, uint32_t seed, void * stream);
/* a plain streaming copy, 16 bytes per lane: the measured-copy yardstick bench.py quotes beside the 8 TB/s spec (count: floats, a multiple of 4) */
int flanhip_copy_dev(const float * d_src, float * d_dst, int64_t count, void * stream);
/* sum of squares of (a - b) and of b, as doubles: d_result[0] = sum (a-b)^2, d_result[1] = sum b^2 */
int flanhip_sqdiff_dev(const float * d_a, const float * d_b, int64_t count, double * d_result, void * stream);

#ifdef __cplusplus
}
#endif
#endif /* FLANHIP_H */
