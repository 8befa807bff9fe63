// pv_kernels.h -- analysis / synthesis kernels of the phase-vocoder hot path (gfx950).
//
// Work decomposition ("chains"): the reference walks the frames of a channel strictly in order because two pieces
// of state cross from frame t-1 to frame t -- the previous phase of every bin in analysis
// (Conversions/AudioPV.cpp:37,44; phase_vocoder.cpp:44-45) and the running phase of every bin in synthesis
// (AudioPV.cpp:105,111; phase_vocoder.cpp:57-59).  Here a channel's frames are cut into chains of L consecutive
// frames; ONE TEAM of threads owns a chain -- a wavefront, or from dft 1024 up a block of 2 to 8 wavefronts sharing the transform
// through LDS -- keeps that state in registers (thread l owns bins l, l+TEAM, l+2 TEAM, ...) and walks its frames in order.
// Chains are independent:
//   analysis : a chain recomputes the phase of frame t0-1 (one extra FFT, no output) to seed `prev`.
//   synthesis: a pre-pass sums the phase increments of every chain (k_phase_sums2), a scan turns the sums into each
//              chain's carry-in (k_phase_scan2), and the overlap-add of the W-hop samples a chain shares with its
//              predecessor goes through a small side buffer that k_ola_fixup adds afterwards, in a fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_device.h"
#include "pv_math.h"

namespace flanhip {

struct MF { float m, f; };


struct AnalyzeParams
	{
	const float * audio;      // [ch][n]
	MF * out;                 // [ch][F][bins]
	const float * window;     // [W]   hann( i/(W-1) )
	const cf * tw;        // [C]   exp(-2 pi i k / C)
	const cf * tw2;       // [C+1] exp(-2 pi i k / N), N = 2C
	int64_t n;                // audio frames per channel
	int64_t F;                // pv frames per channel
	int num_channels;
	int window_size;
	int hop;
	int L;                    // frames per chain
	int chains_per_channel;
	float sample_rate;
	float analysis_rate;
	DivC ar_div;              // analysis_rate as a divisor (pv_math.h)
	double * sums;            // optional [ch][chains][bins]: per-chain sums of the phase increments synthesis will need
	int * nan_out;            // optional: nan_out[0] = nan_epoch when an output MF is NaN/Inf, nan_out[2] = nan_epoch always
	int nan_epoch;            // a fresh non-zero number per launch
	double * group_sums;      // optional (dft 2048 kernel, blocks = groups of 8 chains of one channel): [ch][groups][bins] folded sums of each group's chains
	int groups_per_channel;
	const int * cancel;       // optional: the launching thread's cancel word (core.hip); a wavefront that finds it set when it starts walks no chain
	float * dump;             // 1 KB of device memory nobody reads (the plan's): where lanes whose MFs are not to be written store them instead of branching (pv_kernels_sub.h)
	};

// Is the launch being cancelled?  cancel_peek() issues ONE read past the caches (the word is fine-grained memory) -- at the top of a kernel, so
// that it travels under the prologue's table loads; cancel_seen() consumes it (wave-uniform by construction).
__device__ __forceinline__ int cancel_peek( const int * word )
	{
	return word ? __hip_atomic_load( word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM ) : 0;
	}
__device__ __forceinline__ bool cancel_seen( int peeked ) { return __builtin_amdgcn_readfirstlane( peeked ) != 0; }

// phase_vocoder.cpp:37-52 with the reference's rounding sequence (the file is compiled with -ffp-contract=off); the same helpers as
// the tuned kernels (pv_math.h): atan2_fast (1.8 ulp), divisions by pi2 as exact 3-instruction sequences, |z| as fma + sqrt on
// operands scaled by a power of two.
__device__ __forceinline__ MF phase_vocode_bin( float re, float im, float & prev_phase, float bin_frequency, float expected_phase_diff,
	float analysis_rate, bool use_wrapping )
	{
	const float phase = atan2_fast( im, re );                                         // std::arg
	const float phase_diff = phase - prev_phase;                                      // :44 ( == float( double( phase ) - double( prev ) ): the double difference of two floats is exact, its narrowing the fp32 subtraction's own rounding )
	prev_phase = phase;                                                               // :45
	const float delta_phase = phase_diff - expected_phase_diff;                       // :48
	const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * round_half_away( div_pi2( delta_phase ) ) : delta_phase; // :39-42,49 (roundf, exactly: pv_math.h)
	const float delta_frequency = div_pi2( wrapped * analysis_rate );                 // :50
	MF r;
	r.m = magnitude_scaled( re, im );                                                 // std::abs
	r.f = bin_frequency + delta_frequency;                                            // :52
	return r;
	}

// Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78).  One wavefront per chain, WAVES chains per block.
template<int LOG2C, int WAVES, int T = 1>
__global__ __launch_bounds__( 64 * WAVES * T ) void k_analyze( AnalyzeParams p )
	{
	constexpr int C = 1 << LOG2C;                   // complex points = dft/2
	// A chain is walked by a TEAM of threads: one wavefront (T = 1, WAVES chains per block), or a whole block of T wavefronts sharing
	// one transform through LDS (T > 1, one chain per block: the sizes whose bins do not fit one wavefront's registers).
	static_assert( T == 1 || WAVES == 1, "a team of several wavefronts owns its block" );
	constexpr int TEAM = 64 * T, NT = 64 * WAVES * T;
	constexpr int E = ( C + TEAM - 1 ) / TEAM;      // bins per thread (plus Nyquist on thread 0 of the team)
	// dft 8192 (64 bins per lane): what the smaller sizes keep in registers across frames -- the previous phases -- lives in LDS,
	// the per-bin loops are rolled (4 at a time) and the twiddles stay in global memory (L1): with everything resident and unrolled
	// the kernel spilled 2.4 KB per lane to scratch and ran at a hundredth of the dft 2048 rate.  conversions.hip: analyze_lds_bytes
	// ( BIG and LEAN below describe ONE-WAVEFRONT walks of dft 8192 / 4096+: the launchers give those sizes to teams, T > 1, for which
	// both are off; the variants stay for A/B runs )
	constexpr bool BIG = LOG2C >= 12 && T == 1;
	constexpr int UNR = BIG ? 4 : E;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s_tw = reinterpret_cast<cf*>( smem );                      // [C] (none when BIG)
	float * s_win = reinterpret_cast<float*>( s_tw + ( BIG ? 0 : C ) );      // [W rounded up to even]
	const int wpad = ( p.window_size + 3 ) & ~3;
	cf * s_buf_all = reinterpret_cast<cf*>( s_win + wpad );          // WAVES x padded_len(C)
	float * s_prev_all = reinterpret_cast<float*>( s_buf_all + WAVES * padded_len( C ) );   // WAVES x ( C + 4 ), BIG only

	const int tid = threadIdx.x, lane = T == 1 ? ( tid & 63 ) : tid, wave = T == 1 ? ( tid >> 6 ) : 0;
	// cancellation (core.hip): a team of several wavefronts meets at block barriers in its frame loop, so the decision is the BLOCK's --
	// thread 0 reads the word and everybody learns it at the barrier that ends the prologue
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	if constexpr( !BIG ) for( int i = tid; i < C; i += NT ) s_tw[i] = p.tw[i];
	for( int i = tid; i < p.window_size; i += NT ) s_win[i] = p.window[i];
	__syncthreads();
	if( s_cancel ) return;
	const cf * tw = BIG ? p.tw : s_tw;

	cf * buf = s_buf_all + wave * padded_len( C );
	float * s_prev = s_prev_all + wave * ( C + 4 );
	const int64_t chain = int64_t( blockIdx.x ) * WAVES + wave;
	if( chain >= int64_t( p.chains_per_channel ) * p.num_channels ) return;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const int W = p.window_size, hop = p.hop;
	const int dft = 2 * C;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37

	// per-lane constants: bin frequency (PVBuffer.cpp:443-446), expected phase advance (phase_vocoder.cpp:47), split twiddle.
	// LEAN (dft >= 4096: 32+ bins per lane): only the previous phases stay in registers; the constants are recomputed /
	// fetched (L1) per use and the fused pre-pass sums are not kept (conversions.hip: fused_prepass_supported) -- with all of
	// them resident the dft 8192 kernel spilled 4 KB per lane to scratch.
	constexpr bool LEAN = LOG2C >= 11 && T == 1;
	constexpr int EC = LEAN ? 0 : E;                                          // length of the resident constant arrays
	float binf_r[EC + 1], expect_r[EC + 1], prev[( BIG ? 0 : E ) + 1];
	cf w2_r[EC + 1];
	auto bin_of = [&]( int i ) { return ( i < E ) ? lane + TEAM * i : C; };
	auto binf_of = [&]( int i ) { return LEAN ? float( bin_of( i ) ) * p.sample_rate / float( dft ) : binf_r[LEAN ? 0 : i]; };
	auto expect_of = [&]( int i ) { return LEAN ? binf_of( i ) / p.analysis_rate * FLANHIP_PI2_F : expect_r[LEAN ? 0 : i]; };
	auto w2_of = [&]( int i ) { return LEAN ? p.tw2[min( bin_of( i ), C )] : w2_r[LEAN ? 0 : i]; };
	if constexpr( BIG ) { for( int k = lane; k <= C; k += TEAM ) s_prev[k] = 0.0f; }   // AudioPV.cpp:44
	#pragma unroll
	for( int i = 0; i <= ( BIG ? 0 : E ); ++i )
		{
		prev[i] = 0.0f;                                                       // AudioPV.cpp:44
		if constexpr( !LEAN )
			{
			const int k = bin_of( i );
			binf_r[i] = float( k ) * p.sample_rate / float( dft );
			expect_r[i] = binf_r[i] / p.analysis_rate * FLANHIP_PI2_F;
			if( i < E ) w2_r[i] = p.tw2[min( k, C )];
			}
		}

	double sum[EC + 1];                                                       // fused round trip: see AnalyzeParams::sums
	#pragma unroll
	for( int i = 0; i <= EC; ++i ) sum[i] = 0.0;
	bool bad = false;

	for( int64_t t = ( t0 > 0 ? t0 - 1 : t0 ); t < t1; ++t )
		{
		const bool emit = t >= t0;
		// window the frame into the FFT buffer as C complex points z[i] = ( x[2i], x[2i+1] )  (AudioPV.cpp:52-65)
		const int64_t start = int64_t( hop ) * t - W / 2;
		#pragma unroll UNR
		for( int q = 0; q < E; ++q )
			{
			const int i = lane + TEAM * q;
			if( C >= TEAM || i < C )
				{
				const int s0 = 2 * i, s1 = 2 * i + 1;
				float v0 = 0.0f, v1 = 0.0f;
				if( s0 < W ) { const int64_t a = start + s0; if( a >= 0 && a < p.n ) v0 = x[a] * s_win[s0]; }
				if( s1 < W ) { const int64_t a = start + s1; if( a >= 0 && a < p.n ) v1 = x[a] * s_win[s1]; }
				buf[PAD( i )] = mk( v0, v1 );
				}
			}
		team_sync<TEAM>();
		fft_forward<LOG2C, TEAM>( buf, tw, lane );

		// split the half-size transform into the real transform's bins and phase-vocode each bin (AudioPV.cpp:69-73)
		MF * row = p.out + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		const cf z0 = buf[PAD( 0 )];
		#pragma unroll UNR
		for( int q = 0; q < E; ++q )
			{
			const int k = lane + TEAM * q;
			if( C >= TEAM || k < C )
				{
				const cf zk = buf[PAD( k )];
				const cf zm = buf[PAD( ( C - k ) & ( C - 1 ) )];
				const float ax = 0.5f * ( zk.x + zm.x ), ay = 0.5f * ( zk.y - zm.y );
				const float dx = zk.x - zm.x, dy = zk.y + zm.y;
				const cf w2v = w2_of( q );
				const float c = w2v.x, s = w2v.y;
				float re = ax + 0.5f * __builtin_fmaf( c, dy, s * dx );
				float im = ay - 0.5f * __builtin_fmaf( c, dx, -( s * dy ) );
				if( k == 0 ) { re = z0.x + z0.y; im = 0.0f; }
				MF mf;
				if constexpr( BIG )
					{
					float pr = s_prev[k];
					mf = phase_vocode_bin( re, im, pr, binf_of( q ), expect_of( q ), p.analysis_rate, use_wrapping );
					s_prev[k] = pr;
					}
				else mf = phase_vocode_bin( re, im, prev[BIG ? 0 : q], binf_of( q ), expect_of( q ), p.analysis_rate, use_wrapping );
				if( emit )
					{
					row[k] = mf;
					if constexpr( !LEAN ) sum[q] += double( div_c( mf.f, p.ar_div ) * FLANHIP_PI2_F );      // (the exact constant division: pv_math.h)
					bad |= !( __builtin_fabsf( mf.m ) <= 3.4028235e38f ) || !( __builtin_fabsf( mf.f ) <= 3.4028235e38f );
					}
				}
			}
		if( lane == 0 )
			{
			MF mf;
			if constexpr( BIG )
				{
				float pr = s_prev[C];
				mf = phase_vocode_bin( z0.x - z0.y, 0.0f, pr, binf_of( E ), expect_of( E ), p.analysis_rate, use_wrapping );
				s_prev[C] = pr;
				}
			else mf = phase_vocode_bin( z0.x - z0.y, 0.0f, prev[BIG ? 0 : E], binf_of( E ), expect_of( E ), p.analysis_rate, use_wrapping );
			if( emit )
				{
				row[C] = mf;
				if constexpr( !LEAN ) sum[E] += double( div_c( mf.f, p.ar_div ) * FLANHIP_PI2_F );
				bad |= !( __builtin_fabsf( mf.m ) <= 3.4028235e38f ) || !( __builtin_fabsf( mf.f ) <= 3.4028235e38f );
				}
			}
		team_sync<TEAM>();
		}
	if constexpr( !LEAN ) if( p.sums )
		{
		double * dst = p.sums + chain * ( C + 1 );
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const int k = lane + TEAM * q;
			if( C >= TEAM || k < C ) dst[k] = ( __builtin_fabs( sum[q] ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( sum[q] ) : fold_phase_any( sum[q] );
			}
		if( lane == 0 ) dst[C] = ( __builtin_fabs( sum[E] ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( sum[E] ) : fold_phase_any( sum[E] );
		const bool any_bad = __any( bad );
		if( p.nan_out && ( tid & 63 ) == 0 )                                      // one lane per wavefront of the team
			{
			// no clearing pass: the flag word is "set" when it equals this launch's epoch (written beside it by chain 0)
			if( chain == 0 && lane == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
		}
	}

// -------------------------------------------------------------------------------------------------------------
// synthesis
// -------------------------------------------------------------------------------------------------------------

// phase_vocoder.cpp:57-59:  phase_buffer += float term; if( phase_buffer > pi2 ) phase_buffer = fmod( phase_buffer, pi2 )
__device__ __forceinline__ double fold_phase( double ph )
	{
	if( ph > FLANHIP_PI2_D )
		{
		if( ph < 1.0e6 )
			{
			// exact fmod for moderate quotients: q*pi2 has <= 20+24 significant bits, the difference is exact
			double q = floor( ph / FLANHIP_PI2_D );
			double r = fma( -q, FLANHIP_PI2_D, ph );
			if( r < 0.0 ) r += FLANHIP_PI2_D;
			else if( r >= FLANHIP_PI2_D ) r -= FLANHIP_PI2_D;
			ph = r;
			}
		else ph = fmod( ph, FLANHIP_PI2_D );
		}
	return ph;
	}

__device__ __forceinline__ float phase_term( float f, float analysis_rate )
	{
	return f / analysis_rate * FLANHIP_PI2_F;        // phase_vocoder.cpp:57
	}

struct SynthParams
	{
	const MF * pv;            // [ch][F][bins]
	float * out;              // [ch][F*hop]
	const float * window;     // [W] hann( i/(W-1) ) (unscaled)
	const cf * tw;        // [C]
	const cf * tw2;       // [C+1]
	double * carry;           // [ch][chains][bins]  sums on entry to k_phase_scan2, exclusive carries after
	float * head;             // [ch][chains][W-hop] overlap shared with the previous chain
	int * nan_flag;           // may be null
	int64_t F;
	int64_t out_len;          // F * hop
	int num_channels;
	int window_size;
	int hop;
	int L;
	int chains_per_channel;
	int head_len;             // max( W - hop, 0 )
	int num_bins;
	float analysis_rate;
	float window_scale;       // AudioPV.cpp:99
	DivC ar_div;              // analysis_rate as a divisor (pv_math.h)
	const int * nan_in;       // optional: the { flag, -, epoch } words left by the producer of the PV (fused round trip): set iff equal
	float * dump;             // 512 bytes of workspace that out-of-range lanes of k_synthesize_fast store into
	const double * carry_in;  // optional [ch][bins]: the running phase on entry to frame 0 of this PV (a frame range of a longer one)
	double * total_out;       // optional [ch][bins]: the running phase after the last frame
	int total_only;           // k_phase_scan2 leaves the chain sums untouched (only total_out is produced)
	int * nan_words;          // optional (pre-pass run on behalf of a producer): the producer's { flag, -, epoch } words to write
	int nan_epoch;
	int expect_epoch;          // non-zero: what the host's note says the producer of this workspace left in nan_in[2]; anything else there means ANOTHER producer
	                           // wrote the workspace in between (two callers racing on one workspace): nan_flag |= 2, the output is not to be trusted
	const int * skip_words;   // optional: the pre-pass retires at once when words [4] and [2] agree (the producer of the PV left the sums)
	const int * cancel;        // optional: the launching thread's cancel word (see AnalyzeParams)
	const double * group_sums; // optional (dft 2048 / 4096 team kernels): the producer's per-group sums [ch][groups][bins] -- a scan over THEM (k_phase_scan2<SEG, true>:
	int groups_per_channel;    // an eighth or a quarter of the chains) leaves each group's carry in group_carry, and the synthesis kernel works out its chains'
	double * group_carry;      // carries from that and the chain sums in `carry` (which it leaves untouched): the scan over the chains is not launched
	// k_synthesize_v2 adding the chains' overlaps itself (no k_ola_fixup launch, round 4): one state word per boundary, a side buffer for a
	// chain's LAST partial sums beside `head` (its first ones), this launch's tag
	int * fix_state;           // optional [chains] (team kernels: [chains][wavefronts of a team]): word c belongs to the boundary between chain c - 1 and chain c of a channel
	float * tail;              // [ch][chains][W-hop]
	int fix_tag;               // ( epoch << 2 ): | 1 = the boundary's tail is in `tail`, | 2 = its head is in `head`
	float * ring_ws;           // k_synthesize_mr with MrPlan::ring_ws: [chains][( W + 3 ) & ~3] -- the blocks' overlap-add rings, where they do not fit the LDS
	};

// (the pre-pass kernels k_phase_sums2 / k_phase_scan2 that serve every size live in pv_kernels_fast.h)

// PV::convert_to_audio (Conversions/AudioPV.cpp:86-139).  One wavefront per chain.
template<int LOG2C, int WAVES, int T = 1>
__global__ __launch_bounds__( 64 * WAVES * T ) void k_synthesize( SynthParams p )
	{
	constexpr int C = 1 << LOG2C;
	static_assert( T == 1 || WAVES == 1, "a team of several wavefronts owns its block" );
	constexpr int TEAM = 64 * T, NT = 64 * WAVES * T;    // see k_analyze
	constexpr int E = ( C + TEAM - 1 ) / TEAM;
	// dft 8192: the running phases (fp64, 65 per lane) live in LDS, the per-bin loop is rolled, twiddles come from global memory (L1):
	// see k_analyze.  conversions.hip: synth_lds_bytes
	constexpr bool BIG = LOG2C >= 12 && T == 1;
	constexpr int UNR = BIG ? 2 : E + 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s_tw = reinterpret_cast<cf*>( smem );                       // [C] (none when BIG)
	const int W = p.window_size, hop = p.hop;
	const int wpad = ( W + 3 ) & ~3;
	float * s_win = reinterpret_cast<float*>( s_tw + ( BIG ? 0 : C ) );       // [wpad] scaled window
	cf * s_buf_all = reinterpret_cast<cf*>( s_win + wpad );           // WAVES x padded_len(C+1)
	float * s_ring_all = reinterpret_cast<float*>( s_buf_all + WAVES * padded_len( C + 1 ) ); // WAVES x wpad
	double * s_ph_all = reinterpret_cast<double*>( s_ring_all + WAVES * wpad );   // WAVES x ( C + 2 ), BIG only (wpad is a multiple of 4: 16-byte aligned)

	const int tid = threadIdx.x, lane = T == 1 ? ( tid & 63 ) : tid, wave = T == 1 ? ( tid >> 6 ) : 0;
	__shared__ int s_cancel;                                                    // (see k_analyze)
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	if constexpr( !BIG ) for( int i = tid; i < C; i += NT ) s_tw[i] = p.tw[i];
	const cf * tw = BIG ? p.tw : s_tw;
	for( int i = tid; i < W; i += NT ) s_win[i] = p.window[i] * p.window_scale;   // AudioPV.cpp:102
	float * ring = s_ring_all + wave * wpad;
	for( int i = lane; i < W; i += TEAM ) ring[i] = 0.0f;
	__syncthreads();
	if( s_cancel ) return;

	cf * buf = s_buf_all + wave * padded_len( C + 1 );
	const int64_t chain = int64_t( blockIdx.x ) * WAVES + wave;
	if( chain >= int64_t( p.chains_per_channel ) * p.num_channels ) return;
	const int channel = int( chain / p.chains_per_channel );
	const int chain_in_channel = int( chain % p.chains_per_channel );
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	float * out = p.out + int64_t( channel ) * p.out_len;
	float * head = p.head + chain * p.head_len;
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;                  // first sample this chain touches
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len; // before this: shared with the previous chain

	// running phase (phase_buffer, AudioPV.cpp:105) on entry to the chain
	double ph[( BIG ? 0 : E ) + 1];
	cf w2[BIG ? 1 : E];
	double * s_ph = s_ph_all + wave * ( C + 2 );
	if constexpr( BIG ) { for( int k = lane; k <= C; k += TEAM ) s_ph[k] = p.carry[chain * ( C + 1 ) + k]; }
	else
		{
		#pragma unroll
		for( int i = 0; i <= E; ++i )
			{
			const int k = ( i < E ) ? lane + TEAM * i : C;
			ph[i] = ( ( C >= TEAM || i == E || k < C ) ) ? p.carry[chain * ( C + 1 ) + min( k, C )] : 0.0;
			if( i < E ) w2[i] = p.tw2[min( k, C )];
			}
		}

	int ring_base = 0;                                                         // ring[ring_base] <-> absolute sample `pos`
	int64_t pos = chain_start;
	for( int64_t t = t0; t < t1; ++t )
		{
		// inverse phase vocoder per bin (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61) -> spectrum X[0..C] in LDS
		const MF * row = p.pv + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		#pragma unroll UNR
		for( int q = 0; q <= E; ++q )
			{
			const int k = ( q < E ) ? lane + TEAM * q : C;
			const bool active = ( q < E ) ? ( C >= TEAM || k < C ) : ( lane == 0 );
			if( active )
				{
				const MF mf = row[k];
				// phase_vocoder.cpp:57-60 with the tuned kernels' helpers: f / analysis_rate as an exact division by a constant, the
				// fold without a division, sin / cos as polynomials after a Cody-Waite reduction; the general routines take over for
				// phases outside the range those are exact for (never for a real PV)
				const double term = double( div_c( mf.f, p.ar_div ) * FLANHIP_PI2_F );
				double phase = ( BIG ? s_ph[k] : ph[BIG ? 0 : q] ) + term;
				phase = ( __builtin_fabs( phase ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( phase ) : fold_phase_any( phase );   // (the four-instruction fold, pv_math.h)
				if constexpr( BIG ) s_ph[k] = phase; else ph[BIG ? 0 : q] = phase;
				const float th = float( phase );
				float sn, cs;
				if( __builtin_fabsf( th ) < FLANHIP_SINCOS_FAST_LIMIT ) sincos_fast( th, sn, cs );
				else { const float2 sc = sincos_wide( th ); sn = sc.x; cs = sc.y; }
				buf[PAD( k )] = mk( mf.m * cs, mf.m * sn );           // std::polar
				}
			}
		team_sync<TEAM>();
		// merge X[0..C] into the half-size spectrum Z[k] = A[k] + i B[k]; stored conjugated so that the forward FFT
		// evaluates the inverse transform ( ifft(Z) = conj( fft( conj Z ) ) ).  c2r ignores Im X[0], Im X[C].
		auto merge = [&]( int k, cf xk, cf xm ) -> cf
			{
			if( k == 0 ) { xk.y = 0.0f; xm.y = 0.0f; }
			// A = X[k] + conj X[C-k];  B = ( X[k] - conj X[C-k] ) * exp(+2 pi i k / N)
			const float ax = xk.x + xm.x, ay = xk.y - xm.y;
			const float dx = xk.x - xm.x, dy = xk.y + xm.y;
			const cf w2q = p.tw2[k];
			const float c = w2q.x, s = -w2q.y;                                 // conj of exp(-2 pi i k/N)
			const float bx = __builtin_fmaf( c, dx, -( s * dy ) ), by = __builtin_fmaf( c, dy, s * dx );
			return mk( ax - by, -( ay + bx ) );                                // Z = A + iB = ( ax - by, ay + bx ); stored conjugated
			};
		if constexpr( BIG )
			{
			// in place, a lane owning bin k and its mirror C - k: both are read before either is written, no staging registers
			#pragma unroll 2
			for( int k = lane; k <= C / 2; k += TEAM )
				{
				const cf xk = buf[PAD( k )], xm = buf[PAD( C - k )];
				const cf zk = merge( k, xk, xm );
				if( k != 0 && k != C / 2 ) buf[PAD( C - k )] = merge( C - k, xm, xk );
				buf[PAD( k )] = zk;
				}
			team_sync<TEAM>();
			}
		else
			{
			cf zc[E];
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				const int k = lane + TEAM * q;
				if( C >= TEAM || k < C )
					{
					cf xk = buf[PAD( k )];
					cf xm = buf[PAD( C - k )];
					if( k == 0 ) { xk.y = 0.0f; xm.y = 0.0f; }
					// A = X[k] + conj X[C-k];  B = ( X[k] - conj X[C-k] ) * exp(+2 pi i k / N)
					const float ax = xk.x + xm.x, ay = xk.y - xm.y;
					const float dx = xk.x - xm.x, dy = xk.y + xm.y;
					const cf w2q = BIG ? p.tw2[k] : w2[BIG ? 0 : q];
					const float c = w2q.x, s = -w2q.y;                             // conj of exp(-2 pi i k/N)
					const float bx = __builtin_fmaf( c, dx, -( s * dy ) ), by = __builtin_fmaf( c, dy, s * dx );
					// Z = A + iB = ( ax - by, ay + bx ); store conj
					zc[q] = mk( ax - by, -( ay + bx ) );
					}
				}
			team_sync<TEAM>();
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				const int k = lane + TEAM * q;
				if( C >= TEAM || k < C ) buf[PAD( k )] = zc[q];
				}
			team_sync<TEAM>();
			}
		fft_forward<LOG2C, TEAM>( buf, tw, lane );
		// G = fft( conj Z ):  x[2n] = G[n].x, x[2n+1] = -G[n].y   (AudioPV.cpp:122; samples >= W are discarded)
		// window and accumulate into the ring (AudioPV.cpp:133-134)
		for( int n = lane; 2 * n < W; n += TEAM )
			{
			const cf g = buf[PAD( n )];
			int i0 = ring_base + 2 * n; if( i0 >= W ) i0 -= W;
			ring[i0] += g.x * s_win[2 * n];
			if( 2 * n + 1 < W )
				{
				int i1 = i0 + 1; if( i1 >= W ) i1 -= W;
				ring[i1] += ( -g.y ) * s_win[2 * n + 1];
				}
			}
		team_sync<TEAM>();
		// the oldest `hop` samples are complete as far as this chain is concerned: emit and clear them
		for( int e = lane; e < hop; e += TEAM )
			{
			float v = 0.0f;
			if( e < W )
				{
				int i = ring_base + e; if( i >= W ) i -= W;
				v = ring[i]; ring[i] = 0.0f;
				}
			const int64_t a = pos + e;
			if( a < own_start ) head[a - chain_start] = v;
			else if( a >= 0 && a < p.out_len ) out[a] = v;
			}
		team_sync<TEAM>();
		pos += hop;
		ring_base = ( hop < W ) ? ring_base + hop : 0;
		if( ring_base >= W ) ring_base -= W;
		}
	// flush what is left in the ring (partial sums the next chain's head completes); the last chain of a channel
	// zero-fills up to the end of the output (Audio( format ) is zero-initialised, AudioPV.cpp:95)
	const int64_t ring_end = pos + ( hop < W ? W - hop : 0 );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	for( int64_t a = pos + lane; a < flush_end; a += TEAM )
		{
		float v = 0.0f;
		if( a < ring_end )
			{
			int i = ring_base + int( a - pos ); if( i >= W ) i -= W;
			v = ring[i];
			}
		if( a < own_start ) head[a - chain_start] = v;
		else if( a >= 0 && a < p.out_len ) out[a] = v;
		}
	}

// out[head region of chain c] += head[c]  for every chain c >= 1 of a channel (fixed order => deterministic).
static __global__ __launch_bounds__( 256 ) void k_ola_fixup( SynthParams p )
	{
	const int64_t chain = blockIdx.x;
	const int chain_in_channel = int( chain % p.chains_per_channel );
	if( chain_in_channel == 0 ) return;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t chain_start = int64_t( p.hop ) * ( int64_t( chain_in_channel ) * p.L ) - p.window_size / 2;
	float * out = p.out + int64_t( channel ) * p.out_len;
	const float * head = p.head + chain * p.head_len;
	for( int e = threadIdx.x; e < p.head_len; e += blockDim.x )
		{
		const int64_t a = chain_start + e;
		if( a >= 0 && a < p.out_len ) out[a] += head[e];
		}
	}
// The same sixteen bytes at a time, for the shapes whose heads start on 16-byte boundaries (hop, window / 2, the head length and the channel
// length multiples of 4 samples, 16-byte aligned buffers: every tuned shape): one thread per float4 of every boundary, ( chain, quad ) flat over
// the grid -- a quarter of the memory instructions and no idle first block per channel (8 ch x 60 s: 10 -> 6 us)
static __global__ __launch_bounds__( 256 ) void k_ola_fixup4( SynthParams p )
	{
	typedef float f4 __attribute__(( ext_vector_type( 4 ) ));
	const int quads = p.head_len / 4, per_channel = p.chains_per_channel - 1;      // boundaries of a channel
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= int64_t( p.num_channels ) * per_channel * quads ) return;
	const int q = int( idx % quads );
	const int64_t b = idx / quads;
	const int channel = int( b / per_channel ), chain_in_channel = int( b % per_channel ) + 1;
	const int64_t chain = int64_t( channel ) * p.chains_per_channel + chain_in_channel;
	const int64_t a = int64_t( p.hop ) * ( int64_t( chain_in_channel ) * p.L ) - p.window_size / 2 + 4 * q;
	float * out = p.out + int64_t( channel ) * p.out_len;
	const float * head = p.head + chain * p.head_len + 4 * q;
	if( a >= 0 && a + 3 < p.out_len )
		{
		const f4 h = *reinterpret_cast<const f4*>( head );
		f4 * o = reinterpret_cast<f4*>( out + a );
		*o = *o + h;
		}
	else
		{
		for( int i = 0; i < 4; ++i ) if( a + i >= 0 && a + i < p.out_len ) out[a + i] += head[i];
		}
	}

} // namespace flanhip
