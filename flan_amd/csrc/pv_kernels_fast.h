// pv_kernels_fast.h -- tuned analysis / synthesis kernels for dft 2048 and 4096 (C = 1024 / 2048 complex points).
//
// Same chain decomposition as pv_kernels.h (one wavefront walks L consecutive frames of one channel, state in
// registers), restructured around what limits the generic kernels on gfx950:
//   * the C points of a transform live in REGISTERS in the "natural" layout  z[q] = element (lane + 64 q);
//     the global loads (8-byte, 512 B contiguous per wave instruction), the first FFT pass, the last FFT pass, the
//     per-bin phase-vocoder math, the MF stores and the overlap-add all use that layout, so a frame crosses LDS only
//     between FFT passes (2x) and once for the bin <-> mirror-bin exchange of the real-transform split;
//   * radix schedule 16 x 16 x (C/256): every LDS address is  lane-dependent base + compile-time constant , i.e. one
//     VGPR per access pattern and the constant in the instruction's offset field (the generic kernel spent its
//     VGPRs and half its VALU on address arithmetic);
//   * twiddles come from per-pass tables laid out [r][k] so that lanes read consecutive entries;
//   * divisions by the float constant pi2 are 3 instructions (pv_math.h), |z| is fma + sqrt;
//   * synthesis keeps the overlap-add accumulator in registers (hop a multiple of 128 samples): after each frame the
//     oldest hop samples are final for this chain and leave as coalesced 8-byte stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_device.h"
#include "pv_math.h"
#include "pv_kernels.h"

namespace flanhip {

struct FastTables
	{
	const cf * tw1;       // [15][16]        exp(-2 pi i r k / 256)
	const cf * tw3;       // [R3-1][256]     exp(-2 pi i r j / C)
	const cf * w2;        // [C]             exp(-2 pi i k / 2C)
	};

__host__ __device__ constexpr int fast_tw1_len() { return 15 * 16; }
__host__ __device__ constexpr int fast_tw3_len( int C ) { return ( C / 256 - 1 ) * 256; }

// LDS carve (cf units) shared by both kernels
template<int LOG2C> struct FastLds
	{
	static constexpr int C = 1 << LOG2C;
	static constexpr int TW1 = 0;
	static constexpr int TW3 = TW1 + fast_tw1_len();
	static constexpr int W2 = TW3 + fast_tw3_len( C );
	static constexpr int WIN = W2 + C;                 // window as cf pairs, C entries (2C floats, zero padded)
	static constexpr int BUF = WIN + C;
	static constexpr int BUF_LEN = C + C / 16 + 1;     // highest slot used is PAD( C ) = C + C/16 (synthesis parks X[C] there)
	static constexpr int SUM_LEN = ( C / 64 + 1 ) * 64; // per-wave chain sums (doubles), fused analysis only
	static constexpr int ring_len( int W ) { return ( W + 63 ) & ~63; }         // k_synthesize_fast<.., 0>: a ring of W floats per wave (a whole number of wavefront rows)
	static constexpr size_t ring_bytes( int waves, int W ) { return size_t( waves ) * size_t( ring_len( W ) ) * 4; }
	static constexpr size_t bytes( int waves, bool sums = false ) { return size_t( BUF + waves * BUF_LEN + ( sums && LOG2C < 11 ? waves * SUM_LEN : 0 ) ) * 8; }   // dft 4096 keeps the sums in registers
	};

// diagnostic builds (FLANHIP_STAMPS) pass a functor that reads the clock between the passes; the product passes nothing
struct NoStamp { __device__ __forceinline__ void operator()( int ) const {} };

// ---- the three FFT passes on the register array z[E] (natural layout in, natural layout out) -------------------
// TW_EARLY: request pass 1's twiddles together with its points, ahead of the fence (one LDS round trip less, 30 registers more at the peak)
template<int LOG2C, class ST = NoStamp, bool TW_EARLY = true>
__device__ __forceinline__ void fft_fast( cf ( &z )[( 1 << LOG2C ) / 64], cf * buf, const cf * s_tw1, const cf * s_tw3, int lane, ST && st = ST() )
	{
	constexpr int C = 1 << LOG2C;
	constexpr int E = C / 64;
	constexpr int PER = E / 16;          // radix-16 butterflies per lane in passes 0 and 1
	constexpr int R3 = C / 256;          // radix of the last pass (4 butterflies per lane)
	const int padl = lane + ( lane >> 4 );

	// pass 0: radix 16, sub-transform length 1 -> out[ j*16 + r ], j = lane + 64 b
		{
		cf * wp = buf + 17 * lane;
		#pragma unroll
		for( int b = 0; b < PER; ++b )
			{
			cf v[16];
			#pragma unroll
			for( int r = 0; r < 16; ++r ) v[r] = z[b + PER * r];
			dft_reg<16>( v );
			#pragma unroll
			for( int r = 0; r < 16; ++r ) wp[1088 * b + r] = v[r];
			}
		}
	wave_sync();
	st( 1 );
	// pass 1: radix 16, sub-transform length 16
		{
		cf v[PER][16];
		const cf * rp = buf + padl;
		#pragma unroll
		for( int b = 0; b < PER; ++b )
			{
			#pragma unroll
			for( int r = 0; r < 16; ++r ) v[b][r] = rp[68 * b + 68 * PER * r];
			}
		// the twiddles are requested together with the points, ahead of the fence below (tables are never written after the prologue): one
		// LDS round trip for both instead of two in a row (measured: 1.5 % of either kernel)
		cf tw[15];
		const cf * tp = s_tw1 + ( lane & 15 );
		if constexpr( TW_EARLY )
			{
			#pragma unroll
			for( int r = 1; r < 16; ++r ) tw[r - 1] = tp[( r - 1 ) * 16];
			}
		wave_sync();
		cf * wp = buf + 17 * ( lane & ~15 ) + ( lane & 15 );
		#pragma unroll
		for( int b = 0; b < PER; ++b )
			{
			#pragma unroll
			for( int r = 1; r < 16; ++r ) v[b][r] = cmul( v[b][r], TW_EARLY ? tw[r - 1] : tp[( r - 1 ) * 16] );
			dft_reg<16>( v[b] );
			#pragma unroll
			for( int r = 0; r < 16; ++r ) wp[1088 * b + 17 * r] = v[b][r];
			}
		}
	wave_sync();
	st( 2 );
	// pass 2: radix R3, sub-transform length 256; butterfly j = lane + 64 b, b < 4; result element j + 256 r = lane + 64 (b + 4 r)
		{
		const cf * rp = buf + padl;
		const cf * tp = s_tw3 + lane;
		#pragma unroll
		for( int b = 0; b < 4; ++b )
			{
			cf v[R3];
			#pragma unroll
			for( int r = 0; r < R3; ++r ) v[r] = rp[68 * b + 272 * r];
			#pragma unroll
			for( int r = 1; r < R3; ++r ) v[r] = cmul( v[r], tp[64 * b + ( r - 1 ) * 256] );
			dft_reg<R3>( v );
			#pragma unroll
			for( int r = 0; r < R3; ++r ) z[b + 4 * r] = v[r];
			}
		}
	wave_sync();
	}

__device__ __forceinline__ void load_tables( cf * s, const FastTables & t, const float * window, int W, float scale, int C, int tid, int nthreads )
	{
	const int n1 = fast_tw1_len(), n3 = fast_tw3_len( C );
	for( int i = tid; i < n1; i += nthreads ) s[i] = t.tw1[i];
	for( int i = tid; i < n3; i += nthreads ) s[n1 + i] = t.tw3[i];
	for( int i = tid; i < C; i += nthreads ) s[n1 + n3 + i] = t.w2[i];
	float * win = reinterpret_cast<float*>( s + n1 + n3 + C );
	for( int i = tid; i < 2 * C; i += nthreads ) win[i] = ( i < W ) ? window[i] * scale : 0.0f;
	}

// =================================================================================================================
// Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78)
// =================================================================================================================
template<int LOG2C, int WAVES, bool SUMS>
__global__ __launch_bounds__( 64 * WAVES ) void k_analyze_fast( AnalyzeParams p, FastTables tb )
	{
	using L = FastLds<LOG2C>;
	constexpr int C = 1 << LOG2C;
	constexpr int E = C / 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	load_tables( s, tb, p.window, p.window_size, 1.0f, C, tid, 64 * WAVES );
	__syncthreads();
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const cf * s_w2 = s + L::W2 + lane;
	const cf * s_win = s + L::WIN + lane;
	cf * buf = s + L::BUF + wave * L::BUF_LEN;

	const int64_t chain = int64_t( blockIdx.x ) * WAVES + wave;
	if( chain >= int64_t( p.chains_per_channel ) * p.num_channels || cancel_seen( cancel_peek( p.cancel ) ) ) return;   // (cancelled: core.hip)
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const int W = p.window_size, hop = p.hop;
	const bool w_whole = ( W & 127 ) == 0;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37
	const int padl = lane + ( lane >> 4 );
	const cf * mirror = buf + ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );   // PAD( M - lane ) = 17 M / 16 + this, M % 64 == 0
	using VB = v8f;                                                           // bins evaluated together (pv_math.h)
	constexpr int NV = 8;

	// per-lane constants: bin frequency (PVBuffer.cpp:443-446), expected phase advance (phase_vocoder.cpp:47)
	// (the division by dft, a power of two, is exactly a multiplication; bin_frequency is recomputed per use, 2 instructions)
	const float rdft = 1.0f / float( 2 * C );
	auto bin_frequency = [&]( int q ) { return float( ( q < E ) ? lane + 64 * q : C ) * p.sample_rate * rdft; };
	// ... and so is the expected phase advance (7 instructions) -- registers are the scarce resource of this kernel
	auto expected_advance = [&]( int q ) { return div_c( bin_frequency( q ), p.ar_div ) * FLANHIP_PI2_F; };
	float prev[E + 1];
	#pragma unroll
	for( int q = 0; q <= E; ++q ) prev[q] = 0.0f;                             // AudioPV.cpp:44

	// raw samples of frame t (AudioPV.cpp:52-62) in the natural register layout raw[q] = ( x[2i], x[2i+1] ), i = lane + 64 q.
	// Issued one frame ahead of their use so that the HBM/L2 latency hides under the previous frame's per-bin math.
	// Always exactly E 8-byte loads per lane and no load inside a branch: the number of outstanding memory operations is
	// then static, so the compiler waits with a counted s_waitcnt for these loads only instead of draining the MF
	// stores issued after them.  Frames that stick out of the signal read from clamped addresses and are patched by
	// fix_raw() (pure register arithmetic under a wave-uniform branch).
	struct __attribute__(( packed, aligned( 4 ) )) f2u { float x, y; };      // pair at any 4-byte aligned address
	// (32-bit sample indices: the host routes channels of 2^31 samples or more to the generic kernel)
	const int n32 = int( p.n );
	auto load_raw = [&]( int64_t t, cf ( &raw )[E] )
		{
		const int start = int( int64_t( hop ) * t - W / 2 );
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const int a0c = min( max( start + 2 * ( lane + 64 * q ), 0 ), n32 - 2 );   // n >= 2 on this path (host check)
			const f2u v = *reinterpret_cast<const f2u*>( x + a0c );
			raw[q] = mk( v.x, v.y );
			}
		};
	auto frame_is_interior = [&]( int64_t t )
		{
		const int64_t start = int64_t( hop ) * t - W / 2;
		return w_whole && start >= 0 && start + 2 * int64_t( C ) <= p.n;
		};
	auto fix_raw = [&]( int64_t t, cf ( &raw )[E] )
		{
		const int start = int( int64_t( hop ) * t - W / 2 );
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const int s0 = 2 * ( lane + 64 * q );
			const int a0 = start + s0;
			const int d = a0 - min( max( a0, 0 ), n32 - 2 );                     // 0: pair loaded as is; -1 / +1: shifted by one; else outside
			float v0 = ( d == 0 ) ? raw[q].x : ( d == 1 ? raw[q].y : 0.0f );
			float v1 = ( d == 0 ) ? raw[q].y : ( d == -1 ? raw[q].x : 0.0f );
			if( s0 >= W ) v0 = 0.0f;                                             // AudioPV.cpp:65 (also keeps Inf * 0 out)
			if( s0 + 1 >= W ) v1 = 0.0f;
			raw[q] = mk( v0, v1 );
			}
		};
	// window (AudioPV.cpp:60; the table is zero beyond W, :65), transform, and leave Z both in z[] and (natural order) in
	// buf[] for the mirror reads
	auto transform_frame = [&]( int64_t t, cf ( &z )[E] )
		{
		if( !frame_is_interior( t ) ) fix_raw( t, z );
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const cf w = s_win[64 * q];
			z[q] = mk( z[q].x * w.x, z[q].y * w.y );
			}
		fft_fast<LOG2C>( z, buf, s_tw1, s_tw3, lane );
		#pragma unroll
		for( int q = 0; q < E; ++q ) buf[padl + 68 * q] = z[q];
		wave_sync();
		};

	// bin k = lane + 64 q of the real transform from Z[k] (own register) and Z[C-k] (mirror lane, through LDS)
	auto split_bin = [&]( const cf ( &z )[E], int q, cf z0, float & re, float & im )
		{
		const cf zk = z[q];
		const cf zm = mirror[-68 * q];                                    // k = 0 reads a junk slot, overridden below
		const cf w = s_w2[64 * q];
		const float ax = 0.5f * ( zk.x + zm.x ), ay = 0.5f * ( zk.y - zm.y );
		const float dx = zk.x - zm.x, dy = zk.y + zm.y;
		re = ax + 0.5f * __builtin_fmaf( w.x, dy, w.y * dx );
		im = ay - 0.5f * __builtin_fmaf( w.x, dx, -( w.y * dy ) );
		if( q == 0 ) { re = ( lane == 0 ) ? z0.x + z0.y : re; im = ( lane == 0 ) ? 0.0f : im; }
		};

	// fused round trip: per-chain sums of the phase increments convert_to_audio will integrate (its pre-pass, done here
	// while f is in a register), and a NaN/Inf flag for PVBuffer::is_nan_or_inf
	// The sums live in LDS (double [E+1][64] per wave): 34 more VGPRs would push the kernel into scratch spills, whose
	// reloads count as memory operations and defeat the counted waits above.
	// dft 4096 (one wavefront per SIMD, up to 512 registers with the AGPRs as spill space, and no LDS to spare): the sums are a
	// register array there.
	constexpr bool SUMS_REG = SUMS && LOG2C >= 11;
	double * s_sum = reinterpret_cast<double*>( s + L::BUF + WAVES * L::BUF_LEN ) + wave * L::SUM_LEN + lane;
	double r_sum[SUMS_REG ? E + 1 : 1];
	auto sum_of = [&]( int q ) -> double & { if constexpr( SUMS_REG ) return r_sum[q]; else return s_sum[64 * q]; };
	if constexpr( SUMS )
		{
		#pragma unroll
		for( int q = 0; q <= E; ++q ) sum_of( q ) = 0.0;
		}
	bool bad = false;

	cf z[E], zn[E];
	if( t0 > 0 )
		{
		// halo: only the phases of frame t0-1 are needed (phase_vocoder.cpp:45 leaves them in phase_buffer)
		load_raw( t0 - 1, z );
		transform_frame( t0 - 1, z );
		const cf z0 = buf[0];
		load_raw( t0, zn );
		#pragma unroll
		for( int q0 = 0; q0 < E; q0 += NV )
			{
			VB re, im;
			#pragma unroll
			for( int i = 0; i < NV; ++i ) { float r, m; split_bin( z, q0 + i, z0, r, m ); re[i] = r; im[i] = m; }
			const VB ph2 = atan2_fast_v( im, re );
			#pragma unroll
			for( int i = 0; i < NV; ++i ) prev[q0 + i] = ph2[i];
			}
		prev[E] = atan2_fast( 0.0f, z0.x - z0.y );
		wave_sync();
		}
	else load_raw( t0, zn );

	for( int64_t t = t0; t < t1; ++t )
		{
		#pragma unroll
		for( int q = 0; q < E; ++q ) z[q] = zn[q];
		transform_frame( t, z );
		const cf z0 = buf[0];
		load_raw( min( t + 1, t1 - 1 ), zn );                                  // prefetch (the last frame re-reads itself): in flight during the per-bin math below
		cf * row = reinterpret_cast<cf*>( p.out + ( int64_t( channel ) * p.F + t ) * ( C + 1 ) );
		cf * rowp = row + lane;
		#pragma unroll
		for( int q0 = 0; q0 < E; q0 += NV )
			{
			// NV bins per iteration (k = lane + 64 (q0 + i)), evaluated as one vector stream: NV/2 independent packed
			// instructions per step of every dependent chain (pv_math.h)
			VB re, im, pv, binf;
			#pragma unroll
			for( int i = 0; i < NV; ++i )
				{
				float r, m;
				split_bin( z, q0 + i, z0, r, m );
				re[i] = r; im[i] = m; pv[i] = prev[q0 + i]; binf[i] = bin_frequency( q0 + i );
				}
			// phase_vocoder.cpp:37-52 (AudioPV.cpp:69-73); the dft 2048 kernels' per-bin code (pv_math.h: polar_v shares the reciprocal between
			// atan2 and the magnitude; run-time constants as scalar operands)
			VB phase, m;
			polar_v( re, im, phase, m );
			const VB phase_diff = phase - pv;                                    // == float( double(phase) - double(prev) ), :44
			const VB delta_phase = phase_diff - div_c_each( binf, p.ar_div ) * vsplat<VB>( FLANHIP_PI2_F );    // :47-48
			VB wrapped = delta_phase;
			if( use_wrapping ) wrapped = delta_phase - vsplat<VB>( FLANHIP_PI2_F ) * round_half_away_v( div_pi2_v( delta_phase ) );   // :39-42,49
			VB war;
			#pragma unroll
			for( int i = 0; i < NV; ++i ) war[i] = wrapped[i] * p.analysis_rate;
			const VB f = binf + div_pi2_v( war );                               // :50-52
			#pragma unroll
			for( int i = 0; i < NV; ++i )
				{
				prev[q0 + i] = phase[i];                                         // :45
				__builtin_nontemporal_store( cf{ m[i], f[i] }, rowp + 64 * ( q0 + i ) );   // (written once, read by another kernel later: see k_analyze_v2)
				}
			if constexpr( SUMS )
				{
				const VB term = div_c_each( f, p.ar_div ) * vsplat<VB>( FLANHIP_PI2_F );                       // phase_vocoder.cpp:57-58
				#pragma unroll
				for( int i = 0; i < NV; ++i )
					{
					sum_of( q0 + i ) += double( term[i] );
					bad |= !( __builtin_fabsf( m[i] ) <= 3.4028235e38f ) || !( __builtin_fabsf( f[i] ) <= 3.4028235e38f );
					}
				}
			}
			{
			const float re = z0.x - z0.y;
			const float phase = atan2_fast( 0.0f, re );
			const float phase_diff = float( double( phase ) - double( prev[E] ) );
			prev[E] = phase;
			const float delta_phase = phase_diff - expected_advance( E );
			const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * round_half_away( div_pi2( delta_phase ) ) : delta_phase;
			const float delta_frequency = div_pi2( wrapped * p.analysis_rate );
			const float f = bin_frequency( E ) + delta_frequency;
			row[C] = mk( __builtin_fabsf( re ), f );   // every lane holds the same Nyquist value: an unconditional store keeps the
			                                                    // number of outstanding memory operations static (counted s_waitcnt, no drain)
			if constexpr( SUMS )
				{
				sum_of( E ) += double( div_c( f, p.ar_div ) * FLANHIP_PI2_F );
				bad |= !( __builtin_fabsf( re ) <= 3.4028235e38f ) || !( __builtin_fabsf( f ) <= 3.4028235e38f );
				}
			}
		wave_sync();
		}
	if constexpr( SUMS )
		{
		double * dst = p.sums + chain * ( C + 1 );
		#pragma unroll
		for( int q = 0; q <= E; ++q )
			{
			const double sq = sum_of( q );
			const double v = ( __builtin_fabs( sq ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( sq ) : fold_phase_any( sq );
			if( q < E ) dst[lane + 64 * q] = v;
			else if( lane == 0 ) dst[C] = v;
			}
		const bool any_bad = __any( bad );
		if( p.nan_out && lane == 0 )
			{
			// no clearing pass: the flag word is "set" when it equals this launch's epoch (written beside it by chain 0)
			if( chain == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }   // [4]: the sums of this epoch are in the workspace
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
		}
	}

// =================================================================================================================
// PV::convert_to_audio (Conversions/AudioPV.cpp:86-139); HOPQ = hop / 128 for the hops 128 / 256 / 512 / 1024 (overlap-add
// accumulator in registers); HOPQ = 0: any hop <= window, any window, accumulator = a ring of W floats per wave in LDS
// =================================================================================================================
template<int LOG2C, int WAVES, int HOPQ>
__global__ __launch_bounds__( 64 * WAVES ) void k_synthesize_fast( SynthParams p, FastTables tb )
	{
	using L = FastLds<LOG2C>;
	constexpr int C = 1 << LOG2C;
	constexpr int E = C / 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	load_tables( s, tb, p.window, p.window_size, p.window_scale, C, tid, 64 * WAVES );   // AudioPV.cpp:102: hann * window_scale
	__syncthreads();
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const cf * s_w2 = s + L::W2 + lane;
	const cf * s_win = s + L::WIN + lane;
	cf * buf = s + L::BUF + wave * L::BUF_LEN;

	const int64_t chain = int64_t( blockIdx.x ) * WAVES + wave;
	if( chain >= int64_t( p.chains_per_channel ) * p.num_channels || cancel_seen( cancel_peek( p.cancel ) ) ) return;   // (cancelled: core.hip)
	const int channel = int( chain / p.chains_per_channel );
	const int chain_in_channel = int( chain % p.chains_per_channel );
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	const int W = p.window_size;
	constexpr bool RING = HOPQ == 0;
	const int hop = RING ? p.hop : 128 * HOPQ;
	// register accumulator: everything is in cf units (sample pairs), positions are even because hop, W/2 are multiples of 64;
	// ring accumulator: float units
	float * out1 = p.out + int64_t( channel ) * p.out_len;
	float * head1 = p.head + chain * p.head_len;
	cf * out2 = reinterpret_cast<cf*>( out1 );
	cf * head2 = reinterpret_cast<cf*>( head1 );
	const int rlen = L::ring_len( W );
	float * ring = reinterpret_cast<float*>( s + L::BUF + WAVES * L::BUF_LEN ) + wave * rlen;   // RING only
	if constexpr( RING )
		{
		for( int i = lane; i < rlen; i += 64 ) ring[i] = 0.0f;
		wave_sync();
		}
	int ring_base = 0;                                                         // ring[ring_base] <-> absolute sample `pos`
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;
	const int padl = lane + ( lane >> 4 );
	const cf * mirror = buf + ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );

	double ph[E + 1];                                                          // phase_buffer (AudioPV.cpp:105) on entry to the chain
	#pragma unroll
	for( int q = 0; q <= E; ++q ) ph[q] = p.carry[chain * ( C + 1 ) + ( q < E ? lane + 64 * q : C )];
	cf acc[E];                                                             // overlap-add accumulator: acc[q] <-> samples pos + 128 q + 2 lane (+1)
	#pragma unroll
	for( int q = 0; q < E; ++q ) acc[q] = mk( 0.0f, 0.0f );

	// one 128-sample step of finished (or partial) output leaves the chain
	// Exactly one store instruction per step, never inside a branch (static count of outstanding memory operations, see
	// k_analyze_fast): lanes that fall outside the output are pointed at a 512-byte dump area in the workspace.
	cf * dump2 = reinterpret_cast<cf*>( p.dump ) + lane;
	auto emit_step = [&]( int64_t a0, cf v )
		{
		const int64_t a = a0 + 2 * lane;
#ifndef FLANHIP_STATIC_EMIT   /* measured: the branchy form is 10 % faster than redirecting out-of-range lanes to a dump area */
		(void) dump2;
		if( a0 < own_start ) head2[( a - chain_start ) >> 1] = v;
		else if( a >= 0 && a < p.out_len ) out2[a >> 1] = v;
#else
		cf * dst = ( a0 < own_start ) ? head2 + ( ( a - chain_start ) >> 1 ) : out2 + ( a >> 1 );
		if( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) dst = dump2;
		*dst = v;
#endif
		};

	// MF row of frame t in the natural layout ( m, f ) of bin lane + 64 q; loaded one frame ahead of its use
	auto load_row = [&]( int64_t t, cf ( &mfr )[E], cf & mfny )
		{
		const cf * row = reinterpret_cast<const cf*>( p.pv + ( int64_t( channel ) * p.F + t ) * ( C + 1 ) );
		const cf * rowp = row + lane;
		#pragma unroll
		for( int q = 0; q < E; ++q ) mfr[q] = __builtin_nontemporal_load( rowp + 64 * q );   // the PV is read once
		mfny = __builtin_nontemporal_load( row + C );
		};
	cf mfr[E], mfny;
	load_row( t0, mfr, mfny );

	int64_t pos = chain_start;
	for( int64_t t = t0; t < t1; ++t )
		{
		// ---- inverse phase vocoder per bin (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61)
		cf z[E];
		float mn;
		bool slow = false;
		// f / analysis_rate of the whole row under ONE test of the divisor's plan (see k_synthesize_v2)
		float dv[E + 1];
		if( p.ar_div.exact )
			{
			const float dc = p.ar_div.c, drc = p.ar_div.rc;
			auto div_exact = [&]( float x ) { const float q0 = x * drc; return __builtin_fmaf( __builtin_fmaf( -q0, dc, x ), drc, q0 ); };   // pv_math.h: div_c
			#pragma unroll
			for( int q = 0; q < E; ++q ) dv[q] = div_exact( mfr[q].y );
			dv[E] = div_exact( mfny.y );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) dv[q] = mfr[q].y / p.ar_div.c;
			dv[E] = mfny.y / p.ar_div.c;
			}
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			ph[q] += double( dv[q] * FLANHIP_PI2_F );                            // phase_vocoder.cpp:57-58
			slow |= !( __builtin_fabs( ph[q] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
			z[q].x = mfr[q].x;                                                   // m
			}
		ph[E] += double( dv[E] * FLANHIP_PI2_F );
		slow |= !( __builtin_fabs( ph[E] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
		mn = mfny.x;
#ifndef FLANHIP_STATIC_EMIT
		if( t + 1 < t1 ) load_row( t + 1, mfr, mfny );                          // prefetch: in flight during the transform below
#else
		load_row( min( t + 1, t1 - 1 ), mfr, mfny );                            // prefetch (the last frame re-reads itself): in flight during the transform below
#endif
		cf xn;
		if( __any( slow ) )
			{
			// a phase outside the range the fast helpers are exact for (or a NaN): the general routines for this frame
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				ph[q] = fold_phase_any( ph[q] );
				const float2 sc = sincos_wide( float( ph[q] ) );
				z[q] = mk( z[q].x * sc.y, z[q].x * sc.x );
				}
			ph[E] = fold_phase_any( ph[E] );
			const float2 sc = sincos_wide( float( ph[E] ) );
			xn = mk( mn * sc.y, mn * sc.x );
			}
		else
			{
			#pragma unroll
			for( int q0 = 0; q0 < E; q0 += 4 )
				{
				// four bins per iteration as one vector stream (pv_math.h): two independent packed instructions per step
				v4f th, m4;
				#pragma unroll
				for( int i = 0; i < 4; ++i )
					{
					ph[q0 + i] = fold_phase_loop( ph[q0 + i] );                 // phase_vocoder.cpp:59 (the four-instruction fold, pv_math.h)
					th[i] = float( ph[q0 + i] );
					m4[i] = z[q0 + i].x;
					}
				v4f sn, cs;
				sincos_fast_v( th, sn, cs );
				const v4f xr = m4 * cs, xi = m4 * sn;                            // std::polar, :60
				#pragma unroll
				for( int i = 0; i < 4; ++i ) z[q0 + i] = cf{ xr[i], xi[i] };
				__builtin_amdgcn_sched_barrier( 0 );                            // four bins at a time: keeps the temporaries of 16 bins from overlapping
				}
			ph[E] = fold_phase_loop( ph[E] );
			float sn, cs;
			sincos_fast( float( ph[E] ), sn, cs );
			xn = mk( mn * cs, mn * sn );
			}
		// ---- merge X[0..C] into the half-size spectrum: needs X[k] (own) and X[C-k] (mirror lane): one LDS exchange
		#pragma unroll
		for( int q = 0; q < E; ++q ) buf[padl + 68 * q] = z[q];
		if( lane == 0 ) buf[( C * 17 ) / 16] = xn;                              // slot PAD( C )
		wave_sync();
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			cf xk = z[q];
			cf xm = mirror[-68 * q];                                        // X[ C - k ]; k = 0 pairs with X[C]
			if( q == 0 ) { xk.y = ( lane == 0 ) ? 0.0f : xk.y; xm.y = ( lane == 0 ) ? 0.0f : xm.y; }   // c2r ignores Im X[0], Im X[C]
			const cf w = s_w2[64 * q];
			const float ax = xk.x + xm.x, ay = xk.y - xm.y;                     // A = X[k] + conj X[C-k]
			const float dx = xk.x - xm.x, dy = xk.y + xm.y;                     // D = X[k] - conj X[C-k]
			const float c = w.x, sgn = -w.y;                                    // exp(+2 pi i k / N)
			const float bx = __builtin_fmaf( c, dx, -( sgn * dy ) ), by = __builtin_fmaf( c, dy, sgn * dx );
			z[q] = mk( ax - by, -( ay + bx ) );                        // conj( A + iB ): forward FFT of it = conj of the inverse
			__builtin_amdgcn_sched_barrier( 0 );
			}
		wave_sync();
		fft_fast<LOG2C>( z, buf, s_tw1, s_tw3, lane );
		// ---- G = fft( conj Z ): x[2n] = G[n].x, x[2n+1] = -G[n].y; window and overlap-add (AudioPV.cpp:122-134)
		if constexpr( !RING )
			{
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				const cf w = s_win[64 * q];                                 // zero beyond W
				acc[q].x += z[q].x * w.x;
				acc[q].y += ( -z[q].y ) * w.y;
				}
			#pragma unroll
			for( int q = 0; q < ( RING ? 1 : HOPQ ); ++q ) emit_step( pos + 128 * q, acc[q] );
			#pragma unroll
			for( int q = 0; q < E; ++q ) acc[q] = ( q + HOPQ < E ) ? acc[q + HOPQ] : mk( 0.0f, 0.0f );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				const cf w = s_win[64 * q];
				const int s0 = 2 * ( lane + 64 * q );
				if( s0 < W )
					{
					int i0 = ring_base + s0; if( i0 >= W ) i0 -= W;
					ring[i0] += z[q].x * w.x;
					if( s0 + 1 < W )
						{
						int i1 = i0 + 1; if( i1 >= W ) i1 -= W;
						ring[i1] += ( -z[q].y ) * w.y;
						}
					}
				}
			wave_sync();
			// the oldest `hop` samples are complete as far as this chain is concerned: emit and clear them (hop <= W)
			for( int e = lane; e < hop; e += 64 )
				{
				int i = ring_base + e; if( i >= W ) i -= W;
				const float v = ring[i];
				ring[i] = 0.0f;
				const int64_t a = pos + e;
				if( a < own_start ) head1[a - chain_start] = v;
				else if( a >= 0 && a < p.out_len ) out1[a] = v;
				}
			wave_sync();
			ring_base += hop; if( ring_base >= W ) ring_base -= W;
			}
		pos += hop;
		}
	// flush the partial sums that the next chain's head completes; the last chain zero-fills to the end of the output
	const int64_t ring_end = pos + ( W - hop );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	if constexpr( !RING )
		{
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const int64_t a0 = pos + 128 * q;
			if( a0 < flush_end ) emit_step( a0, acc[q] );
			}
		for( int64_t a0 = pos + 128 * E; a0 < flush_end; a0 += 128 ) emit_step( a0, mk( 0.0f, 0.0f ) );
		}
	else
		for( int64_t a = pos + lane; a < flush_end; a += 64 )
			{
			float v = 0.0f;
			if( a < ring_end ) { int i = ring_base + int( a - pos ); if( i >= W ) i -= W; v = ring[i]; }
			if( a < own_start ) head1[a - chain_start] = v;
			else if( a >= 0 && a < p.out_len ) out1[a] = v;
			}
	}

// ---- pre-pass kernels (all sizes) --------------------------------------------------------------------------------
// One thread per (chain, bin): the chain's phase increments summed in double and folded once (the running phase of
// phase_vocoder.cpp:57-59 modulo pi2); NaN/Inf scan of PVBuffer::is_nan_or_inf.  Streaming read of the PV with 8 rows in
// flight per thread.
static __global__ __launch_bounds__( 256 ) void k_phase_sums2( SynthParams p )
	{
	if( p.skip_words && p.skip_words[4] == p.skip_words[2] && p.skip_words[2] != 0 ) return;   // already there (flanhip_modify_time_dev_fused)
	const int64_t chain = blockIdx.x;
	const int k = blockIdx.y * blockDim.x + threadIdx.x;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int n = int( min( t0 + int64_t( p.L ), p.F ) - t0 );
	bool bad = false;
	if( k < p.num_bins )
		{
		double ph = 0.0;
		const cf * col = reinterpret_cast<const cf*>( p.pv + ( int64_t( channel ) * p.F + t0 ) * p.num_bins + k );
		int i = 0;
		for( ; i + 8 <= n; i += 8 )
			{
			cf v[8];
			#pragma unroll
			for( int u = 0; u < 8; ++u ) v[u] = col[int64_t( i + u ) * p.num_bins];
			#pragma unroll
			for( int u = 0; u < 8; ++u )
				{
				bad |= !( __builtin_fabsf( v[u].x ) <= 3.4028235e38f ) || !( __builtin_fabsf( v[u].y ) <= 3.4028235e38f );
				ph += double( div_c( v[u].y, p.ar_div ) * FLANHIP_PI2_F );
				}
			if( !( __builtin_fabs( ph ) < 1.0e8 ) ) ph = fold_phase_any( ph );       // keep the partial sum small (never for real data)
			}
		for( ; i < n; ++i )
			{
			const cf v = col[int64_t( i ) * p.num_bins];
			bad |= !( __builtin_fabsf( v.x ) <= 3.4028235e38f ) || !( __builtin_fabsf( v.y ) <= 3.4028235e38f );
			ph += double( div_c( v.y, p.ar_div ) * FLANHIP_PI2_F );
			}
		p.carry[chain * p.num_bins + k] = ( __builtin_fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph );
		}
	const bool any_bad = __any( bad );
	if( p.nan_flag && any_bad && ( threadIdx.x & 63 ) == 0 ) atomicOr( p.nan_flag, 1 );
	if( p.nan_words )
		{
		if( blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 ) { p.nan_words[2] = p.nan_epoch; p.nan_words[4] = p.nan_epoch; }
		if( any_bad && ( threadIdx.x & 63 ) == 0 ) p.nan_words[0] = p.nan_epoch;
		}
	}

// The totals of groups of `gsize` consecutive chains from the chains' sums, for a producer that left none (PV::modify_time's and PV::shape's fused forms,
// k_phase_sums2 itself): with them the dft 2048 / 4096 synthesis kernels work out their chains' carries themselves (k_analyze_v2's epilogue does the
// same additions in the same order: run = fold( run + sum of chain w ), w ascending) and the scan over all the chains -- 16 us for 34 MB at config 3 --
// is not launched.  One thread per ( bin, group, channel ): the group's sums are requested together.
template<int GSIZE>
__global__ __launch_bounds__( 256 ) void k_group_sums( SynthParams p, double * out )
	{
	const int bin = blockIdx.x * 256 + threadIdx.x;
	// a handed-over pre-pass is good for ONE convert_to_audio (include/flanhip.h: a second call on the same workspace runs the pre-pass): the word
	// k_phase_sums2 -- launched in front of this kernel -- has just tested is taken back here, as k_phase_scan2 does on the path that runs it
	if( p.skip_words && bin == 0 && blockIdx.y == 0 && blockIdx.z == 0 ) const_cast<int*>( p.skip_words )[4] = 0;
	if( bin >= p.num_bins ) return;
	const int group = blockIdx.y, channel = blockIdx.z;
	const int live = min( GSIZE, p.chains_per_channel - group * GSIZE );
	const double * src = p.carry + ( int64_t( channel ) * p.chains_per_channel + int64_t( group ) * GSIZE ) * p.num_bins + bin;
	double v[GSIZE];
	#pragma unroll
	for( int w = 0; w < GSIZE; ++w ) v[w] = ( w < live ) ? src[int64_t( w ) * p.num_bins] : 0.0;
	double run = 0.0;
	#pragma unroll
	for( int w = 0; w < GSIZE; ++w )
		if( w < live )
			{
			const double t = run + v[w];
			run = ( __builtin_fabs( t ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( t ) : fold_phase_any( t );
			}
	out[( int64_t( channel ) * p.groups_per_channel + group ) * p.num_bins + bin] = run;
	}

// k_phase_sums2 and k_group_sums in one launch, for a convert_to_audio that may or may not have been handed its pre-pass
// (flanhip_synthesize_dev_fused_checked behind PV::modify_time / PV::stretch): when the producer's words say the chain sums are there this is
// k_group_sums; when they do not, a thread first sums its bin over the frames of each of its group's chains -- k_phase_sums2's additions in its
// order, its NaN / Inf scan -- and leaves them in `carry` like that kernel.  The word is NOT taken back here (other blocks are still reading it): the
// synthesis kernel's carry prologue does that (skip_words), one launch later.
template<int GSIZE>
__global__ __launch_bounds__( 256 ) void k_sums_and_groups( SynthParams p, double * out )
	{
	const int bin = blockIdx.x * 256 + threadIdx.x;
	const bool handed = p.skip_words && p.skip_words[4] == p.skip_words[2] && p.skip_words[2] != 0;
	const int group = blockIdx.y, channel = blockIdx.z;
	const int live = min( GSIZE, p.chains_per_channel - group * GSIZE );
	bool bad = false;
	if( bin < p.num_bins )
		{
		double * sums = p.carry + ( int64_t( channel ) * p.chains_per_channel + int64_t( group ) * GSIZE ) * p.num_bins + bin;
		double v[GSIZE];
		if( handed )
			{
			#pragma unroll
			for( int w = 0; w < GSIZE; ++w ) v[w] = ( w < live ) ? sums[int64_t( w ) * p.num_bins] : 0.0;
			}
		else
			{
			for( int w = 0; w < GSIZE; ++w )
				{
				v[w] = 0.0;
				if( w >= live ) continue;
				const int64_t t0 = ( int64_t( group ) * GSIZE + w ) * p.L;
				const int n = int( min( t0 + int64_t( p.L ), p.F ) - t0 );
				const cf * col = reinterpret_cast<const cf*>( p.pv + ( int64_t( channel ) * p.F + t0 ) * p.num_bins + bin );
				double ph = 0.0;
				int i = 0;
				for( ; i + 8 <= n; i += 8 )                                            // (k_phase_sums2, statement for statement)
					{
					cf x[8];
					#pragma unroll
					for( int u = 0; u < 8; ++u ) x[u] = col[int64_t( i + u ) * p.num_bins];
					#pragma unroll
					for( int u = 0; u < 8; ++u )
						{
						bad |= !( __builtin_fabsf( x[u].x ) <= 3.4028235e38f ) || !( __builtin_fabsf( x[u].y ) <= 3.4028235e38f );
						ph += double( div_c( x[u].y, p.ar_div ) * FLANHIP_PI2_F );
						}
					if( !( __builtin_fabs( ph ) < 1.0e8 ) ) ph = fold_phase_any( ph );
					}
				for( ; i < n; ++i )
					{
					const cf x = col[int64_t( i ) * p.num_bins];
					bad |= !( __builtin_fabsf( x.x ) <= 3.4028235e38f ) || !( __builtin_fabsf( x.y ) <= 3.4028235e38f );
					ph += double( div_c( x.y, p.ar_div ) * FLANHIP_PI2_F );
					}
				v[w] = ( __builtin_fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph );
				sums[int64_t( w ) * p.num_bins] = v[w];
				}
			}
		double run = 0.0;
		#pragma unroll
		for( int w = 0; w < GSIZE; ++w )
			if( w < live )
				{
				const double t = run + v[w];
				run = ( __builtin_fabs( t ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( t ) : fold_phase_any( t );
				}
		out[( int64_t( channel ) * p.groups_per_channel + group ) * p.num_bins + bin] = run;
		}
	const bool any_bad = __any( bad );
	if( p.nan_flag && any_bad && ( threadIdx.x & 63 ) == 0 ) atomicOr( p.nan_flag, 1 );
	}

// Exclusive scan of the chain sums along each channel, per bin (modular addition is associative, so the scan is cut in
// SEG segments: 512 / SEG bins x SEG segments per block; each thread sums its segment, the segment totals are scanned through LDS,
// then each thread rewrites its segment as exclusive prefixes).  carry[c] = phase_buffer on entry to chain c.
// SEG = 16 / 32 / 64 (32 / 16 / 8 bins per block), chosen by the host so that a segment has at most 32 chains where it can: few channels mean
// many chains per channel (a stereo minute: 938), and a thread whose segment fits its registers makes ONE trip to memory per pass.
// GROUPS: the same scan over the producer's group totals (group_sums -> group_carry, out of place: a synthesis may be repeated on the same
// workspace), for the synthesis kernels that take their chains' carries from a group's carry and the chain sums themselves.
template<int SEG, bool GROUPS = false>
__global__ __launch_bounds__( 512 ) void k_phase_scan2( SynthParams p )
	{
	constexpr int BINS = 512 / SEG;
	__shared__ double totals[SEG][BINS];
	const int lane_bin = threadIdx.x % BINS, seg = threadIdx.x / BINS;
	const int channel = blockIdx.y;
	const int k = blockIdx.x * BINS + lane_bin;
	if( !GROUPS && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 )
		{
		if( p.nan_in && p.nan_flag && p.nan_in[0] == p.nan_in[2] && p.nan_in[2] != 0 ) atomicOr( p.nan_flag, 1 );
		if( p.expect_epoch && p.nan_in && p.nan_flag && p.nan_in[2] != p.expect_epoch ) atomicOr( p.nan_flag, 2 );   // the sums in this workspace are not the noted producer's
		// the sums become carries below: a handed-over pre-pass is good for one convert_to_audio only
		if( p.skip_words ) const_cast<int*>( p.skip_words )[4] = 0;
		}
	const int n = GROUPS ? p.groups_per_channel : p.chains_per_channel;
	const int seg_len = ( n + SEG - 1 ) / SEG;
	const int i0 = min( seg * seg_len, n ), i1 = min( i0 + seg_len, n );
	const bool live = k < p.num_bins;
	const double * c = ( GROUPS ? p.group_sums : p.carry ) + int64_t( channel ) * n * p.num_bins + ( live ? k : 0 );            // what is scanned ...
	double * d = ( GROUPS ? p.group_carry : p.carry ) + int64_t( channel ) * n * p.num_bins + ( live ? k : 0 );                 // ... and where the exclusive prefixes go
	// x + y folded like phase_vocoder.cpp:59.  The general fold (any magnitude, NaN) calls a routine, and a call inside unrolled code spills
	// everything live; so the unrolled passes use the four-instruction fold (pv_math.h), exact below 3e9 rad, and only note whether any partial
	// sum came within HALF that of its limit -- a thread that saw one (sums of a PV with wildly negative frequencies, NaN) redoes its segment with
	// the general fold in a rolled loop.  (|offset + partial| stays below the limit when both stay below half of it.)
	constexpr double HALF = 0.5 * FLANHIP_FOLD_FAST_LIMIT;
	bool wild = false;
	auto fold_quick = [&]( double v ) { wild |= !( __builtin_fabs( v ) < HALF ); return fold_phase_loop( v ); };
	auto fold_any = []( double v ) { return ( __builtin_fabs( v ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( v ) : fold_phase_any( v ); };
	// The kernel is a handful of blocks waiting for memory: every chain sum a thread needs is requested before the first dependent addition.
	// A segment of up to KEEP chains (any launch with <= 512 chains per channel) stays in registers between the two passes; longer segments
	// (a stereo minute: 938 chains, 59 per segment) go in pieces of KEEP and are read a second time.
	constexpr int KEEP = 32;
	const bool keep = seg_len <= KEEP;
	double held[KEEP];
	double run = 0.0;
	for( int i = i0; i < ( keep ? i0 + 1 : i1 ); i += KEEP )
		{
		#pragma unroll
		for( int u = 0; u < KEEP; ++u ) held[u] = ( live && i + u < i1 ) ? c[int64_t( i + u ) * p.num_bins] : 0.0;
		#pragma unroll
		for( int u = 0; u < KEEP; ++u ) run = fold_quick( run + held[u] );       // + 0.0 past the end: fold( x ) of a folded x is x
		}
	if( wild )
		{
		run = 0.0;
		for( int i = i0; i < i1; ++i ) run = fold_any( run + ( live ? c[int64_t( i ) * p.num_bins] : 0.0 ) );
		}
	totals[seg][lane_bin] = run;
	__syncthreads();
	double offs = ( !GROUPS && p.carry_in && live ) ? p.carry_in[int64_t( channel ) * p.num_bins + k] : 0.0;   // AudioPV.cpp:111 (0 for a whole PV)
	for( int s2 = 0; s2 < seg; ++s2 ) offs = fold_any( offs + totals[s2][lane_bin] );
	run = offs;
	if( !GROUPS && p.total_only )
		{
		if( p.total_out && live && seg == SEG - 1 ) p.total_out[int64_t( channel ) * p.num_bins + k] = fold_any( offs + totals[seg][lane_bin] );
		return;
		}
	wild |= !( __builtin_fabs( offs ) < HALF );
	if( wild )
		{
		for( int i = i0; i < i1; ++i )
			{
			const double v = live ? c[int64_t( i ) * p.num_bins] : 0.0;
			if( live ) d[int64_t( i ) * p.num_bins] = run;
			run = fold_any( run + v );
			}
		}
	else
		for( int i = i0; i < ( keep ? i0 + 1 : i1 ); i += KEEP )
			{
			if( !keep )
				{
				#pragma unroll
				for( int u = 0; u < KEEP; ++u ) held[u] = ( live && i + u < i1 ) ? c[int64_t( i + u ) * p.num_bins] : 0.0;
				}
			#pragma unroll
			for( int u = 0; u < KEEP; ++u )
				{
				if( live && i + u < i1 ) d[int64_t( i + u ) * p.num_bins] = run;
				run = fold_phase_loop( run + held[u] );
				}
			}
	if( !GROUPS && p.total_out && live && seg == SEG - 1 ) p.total_out[int64_t( channel ) * p.num_bins + k] = run;   // the running phase after the last chain
	}

} // namespace flanhip
