// flanhip_internal.h -- host-side plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "flanhip.h"
#include "cf_type.h"

namespace flanhip {

void set_error( const char * fmt, ... );

#define FLANHIP_CHECK( expr )                                                                         \
	do { hipError_t e_ = ( expr );                                                                    \
		if( e_ != hipSuccess ) {                                                                      \
			::flanhip::set_error( "%s failed: %s (%s:%d)", #expr, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
			return FLANHIP_ERR_HIP; } } while( 0 )

#define FLANHIP_REQUIRE( cond, code, msg )                                                            \
	do { if( !( cond ) ) { ::flanhip::set_error( "%s: %s", __func__, msg ); return code; } } while( 0 )

int require_device();   // FLANHIP_OK or FLANHIP_ERR_NO_DEVICE

inline bool is_pow2( int64_t n ) { return n > 0 && ( n & ( n - 1 ) ) == 0; }
inline int ilog2( int64_t n ) { int l = 0; while( ( int64_t( 1 ) << l ) < n ) ++l; return l; }
inline bool cancelled( volatile int * c ) { return c && *c != 0; }
// cancellation inside a launch (core.hip): the calling thread's cancel word on the current device (kernels poll it; nullptr if it cannot be had),
// and a wait for a stream that polls `poll( user )` and raises the word when it says so (FLANHIP_ERR_CANCELLED then)
int * thread_cancel_word( hipStream_t s );
int wait_cancellable( hipStream_t s, int ( *poll )( void * ), void * user );
inline int poll_volatile_int( void * user ) { volatile int * c = static_cast<volatile int*>( user ); return c && *c != 0; }

// Device tables for one (window, dft) pair on one device: the analogue of the reference's FFTHelper plan
// (FFTHelper.cpp:16-26) plus the sampled Hann window (AudioPV.cpp:30-34).  Built once, cached (core.hip: get_plan).
struct Plan
	{
	float * d_window = nullptr;    // [W]   hann( i/(W-1) ), WindowFunctions.cpp:10-13 evaluated on the host in double
	cf * d_tw = nullptr;       // [C]   exp(-2 pi i k / C)
	cf * d_tw2 = nullptr;      // [C+1] exp(-2 pi i k / (2C))
	cf * d_tw1f = nullptr;     // fast path (dft 2048/4096): [15][16]      exp(-2 pi i r k / 256)
	cf * d_tw3f = nullptr;     // fast path:                 [C/256-1][256] exp(-2 pi i r j / C)
	cf * d_bs_tw = nullptr;    // chirp-z sizes (bs_plan.h): [M] exp(-2 pi i j / M)
	d2 * d_bs_chirp = nullptr; //                            [C] exp(+pi i n^2 / C), in double
	d2 * d_bs_bh = nullptr;    //                            [M] the chirp's transform / M, in double
	float * d_dump = nullptr;  // 1 KB nobody reads: the target of stores that must be issued but not land (AnalyzeParams::dump)
	cf * d_team = nullptr;     // dft 8192 / 16384 (pv_kernels_team.h, R = dft / 2048): tw1 [240], tw3 [768], twj [(R-1) 512], tws [(R/2) 512], two [2 R], back to back
	};
// offsets (in cf) of the team kernels' tables inside Plan::d_team
inline int team_radix( int dft_size ) { return dft_size == 8192 ? 4 : dft_size == 16384 ? 8 : 0; }
struct TeamTableLayout { int tw1, tw3, twj, tws, two, total; };
inline TeamTableLayout team_table_layout( int R ) { TeamTableLayout l; l.tw1 = 0; l.tw3 = 240; l.twj = 1008; l.tws = l.twj + ( R - 1 ) * 512; l.two = l.tws + ( R / 2 ) * 512; l.total = l.two + 2 * R; return l; }
struct PlanRef { Plan plan; PlanRef() = default; PlanRef( const PlanRef & ) = delete; PlanRef & operator=( const PlanRef & ) = delete; ~PlanRef(); };   // owns the tables
int get_plan( int window_size, int dft_size, std::shared_ptr<const PlanRef> * out );   // keep the reference until the kernels that read the tables are launched

// Division by a run-time constant: { c, RN(1/c), exact } where `exact` says the 3-instruction quotient (pv_math.h div_c) was
// checked on the device against the hardware division for every float |x| >= 1e-30.  Cached per value of c.
struct DivPlan { float c, rc; int exact; };
int get_div_plan( float c, DivPlan * out );

// Chain length heuristics (frames per wavefront-chain)
int choose_chain_length( int64_t num_channels, int64_t num_frames, int min_len, int target_chains, int group = 1 );   // group: chains per block of the kernels (core.hip)
// flanhip_debug_option (include/flanhip.h): per-thread test / A-B hooks, all off by default
struct DebugOptions
	{
	int chain_len = 0;        // FLANHIP_DEBUG_CHAIN_LEN: frames per chain (0: the library's choice)
	int target_chains = 0;    // FLANHIP_DEBUG_TARGET_CHAINS: chains the tuned kernels are cut for (0: what the device holds at once)
	int force_generic = 0;    // FLANHIP_DEBUG_FORCE_GENERIC: never take the tuned dft 2048 / 4096 kernels (A/B and parity of both paths)
	int no_fast_div = 0;      // FLANHIP_DEBUG_NO_FAST_DIV: hardware division by the analysis rate even where the 3-instruction quotient is proven
	int ana_variant = 0, syn_variant = 0;      // dft 2048: ablated instantiations of diagnostic builds; synthesis 2 = behind the scan kernel
	int ana11_old = 0, syn11_old = 0;          // dft 4096: 1 = the round-1 kernels instead of the team kernels (A/B predecessor)
	int resample_direct = 0;                   // 1: the 2:1 block convolver always as direct sums; 2: its radix-16 generation (k_resample_ols2)
	int wide_offsets = 0;                      // 1: 64-bit element offsets where a kernel would choose 32-bit ones by the grid's size (k_stretch_map)
	int inline_fixup = 0;                      // the dft 2048 synthesis kernel adding the chains' overlaps itself instead of k_ola_fixup in a launch of its own: 0 where its chains are long enough (the library's choice), 1 always, 2 never
	int no_sub = 0;                            // 1: dft 512 / 256 never on the several-chains-per-wavefront kernels (pv_kernels_sub.h): the one-wavefront kernels of pv_kernels_v3.h / the generic ones (A/B)
	int force_direct = 0;                      // 1: dft sizes without power-of-two kernels as direct fp64 sums (pv_kernels_any.h), never the mixed-radix kernels
	};
DebugOptions & debug_options();
bool force_generic();
int cu_count();         // compute units of the current device

// Launchers implemented in analyze.hip / synthesize.hip / processors.hip
// d_fused_ws (optional): a synthesis workspace for the PV being produced; analysis leaves the per-chain phase sums and a NaN flag
// there so that launch_synthesize( ..., presummed = true ) can skip its pre-pass.
int launch_analyze( const float * d_audio, int64_t ch, int64_t n, float sr, int W, int hop, int dft, flanhip_MF * d_out, void * d_fused_ws, hipStream_t s );
// presummed: 0 = run the pre-pass; 1 = the workspace holds the chain sums (left by the analysis that produced the PV);
// 2 = it may hold them (left by modify_time when its time map allowed it): the pre-pass is launched and retires at once if so;
// 3 = it holds them and no producer words (flanhip_synthesize_prepass_dev)
// carry_in / total_out (optional, double[ch][bins]): the running phases on entry to / after this PV, for a PV that is a frame
// range of a longer one (flanhip_synthesize_prepass_dev / flanhip_synthesize_dev_carry); prepass_only: stop after the pre-pass
int launch_synthesize( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W, float * d_out,
	void * d_ws, int * d_nan, int presummed, hipStream_t s, const double * d_carry_in = nullptr, double * d_total_out = nullptr,
	bool prepass_only = false, int stage_mask = 0xF );     // stage_mask: 1 sums, 2 scan, 4 main, 8 fix-up (flanhip_synthesize_dev_stages)

// A fresh non-zero number per producer launch: workspace words are "set" when they equal the launch's epoch (no clearing pass).
int next_epoch();

// groups: aligned runs of 8 consecutive chains of a channel (one 8-wave block of the dft 2048 kernels); group_bytes: one fp64 sum per
// (channel, group, bin) behind the 1024-byte tail -- what lets the synthesis kernel compute its own carries (no scan kernel)
struct SynthLayout { int hop, dft, L, chains_per_channel, head_len, groups_per_channel; size_t carry_bytes, head_bytes, group_offset, group_bytes, total_bytes;
	size_t flags_offset;              // bins + 1 words for a producer's own notes (PV::modify_time: which columns of the time map run backwards)
	size_t fix_offset, tail_offset;   // the synthesis kernels' own overlap fix-up: eight state words per chain (one per wavefront of a team), then a second side buffer the size of the heads' (0: not this shape)
	bool any; size_t any_spec_offset, any_frames_offset;      // any: a dft size without FFT kernels (pv_kernels_any.h) and its scratch in the workspace
	size_t mr_ring_offset;            // the mixed-radix synthesis' rings in the workspace (0: in LDS)
	size_t bsg_offset;                // the chirp-z kernels' buffers and state in the workspace (BsPlan::glob; 0: in LDS)
	bool big; size_t big_out_offset, big_head_offset, big_ring_offset; };      // big: a dft size above 16384 served by pv_kernels_big.h; its units' partial output streams and heads
// which producer last left its pre-pass in a synthesis workspace (host-side note, keyed by the workspace pointer, written when the
// producer is launched and read when flanhip_synthesize_dev_fused is): 1 = chain sums AND group sums (the dft 2048 analysis kernel)
void note_workspace_producer( const void * d_ws, int kind, int epoch = 0 );   // epoch: what the producer's kernel leaves in the workspace's tag word [2]
int workspace_producer( const void * d_ws, int * epoch = nullptr );
int synth_layout( int64_t ch, int64_t F, int bins, float sr, float ar, int W, SynthLayout * out );

// Interpolators: a named kind (FLANHIP_INTERP_LINEAR .. _SINE) or a live table registered with flanhip_interp_table_create (processors_ext.hip)
bool valid_interp( int kind );
float interp_eval_host( int kind, float x );            // one value on the host (get_frame): named kinds like Utility/Interpolator.cpp, tables like the device
// every translation unit with kernels that call interpolate() keeps its own copy of the table pointers
int processors_set_interp_lut( int slot, const float * d_table );
int processors_ext_set_interp_lut( int slot, const float * d_table );
int processors_arrange_set_interp_lut( int slot, const float * d_table );

} // namespace flanhip
