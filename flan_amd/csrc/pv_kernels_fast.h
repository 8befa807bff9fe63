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
	const float2 * tw1;       // [15][16]        exp(-2 pi i r k / 256)
	const float2 * tw3;       // [R3-1][256]     exp(-2 pi i r j / C)
	const float2 * w2;        // [C]             exp(-2 pi i k / 2C)
	};

__host__ __device__ constexpr int fast_tw1_len() { return 15 * 16; }
__host__ __device__ constexpr int fast_tw3_len( int C ) { return ( C / 256 - 1 ) * 256; }

// LDS carve (float2 units) shared by both kernels
template<int LOG2C> struct FastLds
	{
	static constexpr int C = 1 << LOG2C;
	static constexpr int TW1 = 0;
	static constexpr int TW3 = TW1 + fast_tw1_len();
	static constexpr int W2 = TW3 + fast_tw3_len( C );
	static constexpr int WIN = W2 + C;                 // window as float2 pairs, C entries (2C floats, zero padded)
	static constexpr int BUF = WIN + C;
	static constexpr int BUF_LEN = padded_len( C + 1 );
	static constexpr size_t bytes( int waves ) { return size_t( BUF + waves * BUF_LEN ) * 8; }
	};

// ---- the three FFT passes on the register array z[E] (natural layout in, natural layout out) -------------------
template<int LOG2C>
__device__ __forceinline__ void fft_fast( float2 ( &z )[( 1 << LOG2C ) / 64], float2 * buf, const float2 * s_tw1, const float2 * s_tw3, int lane )
	{
	constexpr int C = 1 << LOG2C;
	constexpr int E = C / 64;
	constexpr int PER = E / 16;          // radix-16 butterflies per lane in passes 0 and 1
	constexpr int R3 = C / 256;          // radix of the last pass (4 butterflies per lane)
	const int padl = lane + ( lane >> 4 );

	// pass 0: radix 16, sub-transform length 1 -> out[ j*16 + r ], j = lane + 64 b
		{
		float2 * wp = buf + 17 * lane;
		#pragma unroll
		for( int b = 0; b < PER; ++b )
			{
			float2 v[16];
			#pragma unroll
			for( int r = 0; r < 16; ++r ) v[r] = z[b + PER * r];
			dft_reg<16>( v );
			#pragma unroll
			for( int r = 0; r < 16; ++r ) wp[1088 * b + r] = v[r];
			}
		}
	wave_sync();
	// pass 1: radix 16, sub-transform length 16
		{
		float2 v[PER][16];
		const float2 * rp = buf + padl;
		#pragma unroll
		for( int b = 0; b < PER; ++b )
			{
			#pragma unroll
			for( int r = 0; r < 16; ++r ) v[b][r] = rp[68 * b + 68 * PER * r];
			}
		wave_sync();
		const float2 * tp = s_tw1 + ( lane & 15 );
		float2 * wp = buf + 17 * ( lane & ~15 ) + ( lane & 15 );
		#pragma unroll
		for( int b = 0; b < PER; ++b )
			{
			#pragma unroll
			for( int r = 1; r < 16; ++r ) v[b][r] = cmul( v[b][r], tp[( r - 1 ) * 16] );
			dft_reg<16>( v[b] );
			#pragma unroll
			for( int r = 0; r < 16; ++r ) wp[1088 * b + 17 * r] = v[b][r];
			}
		}
	wave_sync();
	// pass 2: radix R3, sub-transform length 256; butterfly j = lane + 64 b, b < 4; result element j + 256 r = lane + 64 (b + 4 r)
		{
		const float2 * rp = buf + padl;
		const float2 * tp = s_tw3 + lane;
		#pragma unroll
		for( int b = 0; b < 4; ++b )
			{
			float2 v[R3];
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

__device__ __forceinline__ void load_tables( float2 * s, const FastTables & t, const float * window, int W, float scale, int C, int tid, int nthreads )
	{
	const int n1 = fast_tw1_len(), n3 = fast_tw3_len( C );
	for( int i = tid; i < n1; i += nthreads ) s[i] = t.tw1[i];
	for( int i = tid; i < n3; i += nthreads ) s[n1 + i] = t.tw3[i];
	for( int i = tid; i < C; i += nthreads ) s[n1 + n3 + i] = t.w2[i];
	float * win = reinterpret_cast<float*>( s + n1 + n3 + C );
	for( int i = tid; i < 2 * C; i += nthreads ) win[i] = ( i < W ) ? window[i] * scale : 0.0f;
	}

// magnitude with the operands pre-scaled by a power of two (exact), so that the squares neither overflow nor underflow
// for any finite input: |z| = 2^e * sqrt( (re 2^-e)^2 + (im 2^-e)^2 ), e = exponent of max(|re|,|im|).
__device__ __forceinline__ float magnitude_scaled( float re, float im )
	{
	const float a = __builtin_fmaxf( __builtin_fabsf( re ), __builtin_fabsf( im ) );
	const int e = __builtin_amdgcn_frexp_expf( a );
	const float rs = __builtin_ldexpf( re, -e ), is = __builtin_ldexpf( im, -e );
	return __builtin_ldexpf( __builtin_amdgcn_sqrtf( __builtin_fmaf( rs, rs, is * is ) ), e );
	}

// =================================================================================================================
// Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78)
// =================================================================================================================
template<int LOG2C, int WAVES>
__global__ __launch_bounds__( 64 * WAVES ) void k_analyze_fast( AnalyzeParams p, FastTables tb )
	{
	using L = FastLds<LOG2C>;
	constexpr int C = 1 << LOG2C;
	constexpr int E = C / 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float2 * s = reinterpret_cast<float2*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	load_tables( s, tb, p.window, p.window_size, 1.0f, C, tid, 64 * WAVES );
	__syncthreads();
	const float2 * s_tw1 = s + L::TW1;
	const float2 * s_tw3 = s + L::TW3;
	const float2 * s_w2 = s + L::W2 + lane;
	const float2 * s_win = s + L::WIN + lane;
	float2 * buf = s + L::BUF + wave * L::BUF_LEN;

	const int64_t chain = int64_t( blockIdx.x ) * WAVES + wave;
	if( chain >= int64_t( p.chains_per_channel ) * p.num_channels ) return;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const int W = p.window_size, hop = p.hop;
	const int WQ = ( W + 127 ) >> 7;                                         // 128-sample steps the window covers
	const bool w_whole = ( W & 127 ) == 0;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37
	const int padl = lane + ( lane >> 4 );
	const float2 * mirror = buf + ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );   // PAD( M - lane ) = 17 M / 16 + this, M % 64 == 0

	// per-lane constants: bin frequency (PVBuffer.cpp:443-446), expected phase advance (phase_vocoder.cpp:47)
	float binf[E + 1], expect[E + 1], prev[E + 1];
	#pragma unroll
	for( int q = 0; q <= E; ++q )
		{
		const int k = ( q < E ) ? lane + 64 * q : C;
		binf[q] = float( k ) * p.sample_rate / float( 2 * C );
		expect[q] = binf[q] / p.analysis_rate * FLANHIP_PI2_F;
		prev[q] = 0.0f;                                                       // AudioPV.cpp:44
		}

	// raw samples of frame t (AudioPV.cpp:52-62) in the natural register layout raw[q] = ( x[2i], x[2i+1] ), i = lane + 64 q.
	// Issued one frame ahead of their use so that the HBM/L2 latency hides under the previous frame's per-bin math.
	auto load_raw = [&]( int64_t t, float2 ( &raw )[E] )
		{
		const int64_t start = int64_t( hop ) * t - W / 2;
		const float * xs = x + start;
		const bool interior = w_whole && start >= 0 && start + W <= p.n && ( ( reinterpret_cast<uintptr_t>( xs ) & 7 ) == 0 );
		if( interior )
			{
			const float2 * xp = reinterpret_cast<const float2*>( xs ) + lane;
			#pragma unroll
			for( int q = 0; q < E; ++q ) raw[q] = ( q < WQ ) ? xp[64 * q] : make_float2( 0.0f, 0.0f );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				const int s0 = 2 * ( lane + 64 * q );
				float v0 = 0.0f, v1 = 0.0f;
				if( s0 < W )     { const int64_t a = start + s0;     if( a >= 0 && a < p.n ) v0 = x[a]; }
				if( s0 + 1 < W ) { const int64_t a = start + s0 + 1; if( a >= 0 && a < p.n ) v1 = x[a]; }
				raw[q] = make_float2( v0, v1 );
				}
			}
		};
	// window (AudioPV.cpp:60; the table is zero beyond W, :65), transform, and leave Z both in z[] and (natural order) in
	// buf[] for the mirror reads
	auto transform_frame = [&]( float2 ( &z )[E] )
		{
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const float2 w = s_win[64 * q];
			z[q] = make_float2( z[q].x * w.x, z[q].y * w.y );
			}
		fft_fast<LOG2C>( z, buf, s_tw1, s_tw3, lane );
		#pragma unroll
		for( int q = 0; q < E; ++q ) buf[padl + 68 * q] = z[q];
		wave_sync();
		};

	// bin k = lane + 64 q of the real transform from Z[k] (own register) and Z[C-k] (mirror lane, through LDS)
	auto split_bin = [&]( const float2 ( &z )[E], int q, float2 z0, float & re, float & im )
		{
		const float2 zk = z[q];
		const float2 zm = mirror[-68 * q];                                    // k = 0 reads a junk slot, overridden below
		const float2 w = s_w2[64 * q];
		const float ax = 0.5f * ( zk.x + zm.x ), ay = 0.5f * ( zk.y - zm.y );
		const float dx = zk.x - zm.x, dy = zk.y + zm.y;
		re = ax + 0.5f * __builtin_fmaf( w.x, dy, w.y * dx );
		im = ay - 0.5f * __builtin_fmaf( w.x, dx, -( w.y * dy ) );
		if( q == 0 ) { re = ( lane == 0 ) ? z0.x + z0.y : re; im = ( lane == 0 ) ? 0.0f : im; }
		};

	float2 z[E], zn[E];
	if( t0 > 0 )
		{
		// halo: only the phases of frame t0-1 are needed (phase_vocoder.cpp:45 leaves them in phase_buffer)
		load_raw( t0 - 1, z );
		transform_frame( z );
		const float2 z0 = buf[0];
		load_raw( t0, zn );
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			float re, im;
			split_bin( z, q, z0, re, im );
			prev[q] = atan2_fast( im, re );
			}
		prev[E] = atan2_fast( 0.0f, z0.x - z0.y );
		wave_sync();
		}
	else load_raw( t0, zn );

	for( int64_t t = t0; t < t1; ++t )
		{
		#pragma unroll
		for( int q = 0; q < E; ++q ) z[q] = zn[q];
		transform_frame( z );
		const float2 z0 = buf[0];
		if( t + 1 < t1 ) load_raw( t + 1, zn );                                // prefetch: in flight during the per-bin math below
		float2 * row = reinterpret_cast<float2*>( p.out + ( int64_t( channel ) * p.F + t ) * ( C + 1 ) );
		float2 * rowp = row + lane;
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			float re, im;
			split_bin( z, q, z0, re, im );
			// phase_vocoder.cpp:37-52 (AudioPV.cpp:69-73)
			const float phase = atan2_fast( im, re );
			const float phase_diff = float( double( phase ) - double( prev[q] ) );
			prev[q] = phase;
			const float delta_phase = phase_diff - expect[q];
			const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * roundf( div_pi2( delta_phase ) ) : delta_phase;
			const float delta_frequency = div_pi2( wrapped * p.analysis_rate );
			rowp[64 * q] = make_float2( magnitude_scaled( re, im ), binf[q] + delta_frequency );
			}
			{
			const float re = z0.x - z0.y;
			const float phase = atan2_fast( 0.0f, re );
			const float phase_diff = float( double( phase ) - double( prev[E] ) );
			prev[E] = phase;
			const float delta_phase = phase_diff - expect[E];
			const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * roundf( div_pi2( delta_phase ) ) : delta_phase;
			const float delta_frequency = div_pi2( wrapped * p.analysis_rate );
			if( lane == 0 ) row[C] = make_float2( __builtin_fabsf( re ), binf[E] + delta_frequency );
			}
		wave_sync();
		}
	}

// =================================================================================================================
// PV::convert_to_audio (Conversions/AudioPV.cpp:86-139); HOPQ = hop / 128
// =================================================================================================================
template<int LOG2C, int WAVES, int HOPQ>
__global__ __launch_bounds__( 64 * WAVES ) void k_synthesize_fast( SynthParams p, FastTables tb )
	{
	using L = FastLds<LOG2C>;
	constexpr int C = 1 << LOG2C;
	constexpr int E = C / 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float2 * s = reinterpret_cast<float2*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	load_tables( s, tb, p.window, p.window_size, p.window_scale, C, tid, 64 * WAVES );   // AudioPV.cpp:102: hann * window_scale
	__syncthreads();
	const float2 * s_tw1 = s + L::TW1;
	const float2 * s_tw3 = s + L::TW3;
	const float2 * s_w2 = s + L::W2 + lane;
	const float2 * s_win = s + L::WIN + lane;
	float2 * buf = s + L::BUF + wave * L::BUF_LEN;

	const int64_t chain = int64_t( blockIdx.x ) * WAVES + wave;
	if( chain >= int64_t( p.chains_per_channel ) * p.num_channels ) return;
	const int channel = int( chain / p.chains_per_channel );
	const int chain_in_channel = int( chain % p.chains_per_channel );
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	const int W = p.window_size;
	constexpr int hop = 128 * HOPQ;
	// everything below is in float2 units (sample pairs): positions are even because hop, W/2... are multiples of 64
	float2 * out2 = reinterpret_cast<float2*>( p.out + int64_t( channel ) * p.out_len );
	float2 * head2 = reinterpret_cast<float2*>( p.head + chain * p.head_len );
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;
	const int padl = lane + ( lane >> 4 );
	const float2 * mirror = buf + ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );

	double ph[E + 1];                                                          // phase_buffer (AudioPV.cpp:105) on entry to the chain
	#pragma unroll
	for( int q = 0; q <= E; ++q ) ph[q] = p.carry[chain * ( C + 1 ) + ( q < E ? lane + 64 * q : C )];
	float2 acc[E];                                                             // overlap-add accumulator: acc[q] <-> samples pos + 128 q + 2 lane (+1)
	#pragma unroll
	for( int q = 0; q < E; ++q ) acc[q] = make_float2( 0.0f, 0.0f );

	// one 128-sample step of finished (or partial) output leaves the chain
	auto emit_step = [&]( int64_t a0, float2 v )
		{
		const int64_t a = a0 + 2 * lane;
		if( a0 < own_start ) head2[( a - chain_start ) >> 1] = v;
		else if( a >= 0 && a < p.out_len ) out2[a >> 1] = v;
		};

	// MF row of frame t in the natural layout ( m, f ) of bin lane + 64 q; loaded one frame ahead of its use
	auto load_row = [&]( int64_t t, float2 ( &mfr )[E], float2 & mfny )
		{
		const float2 * row = reinterpret_cast<const float2*>( p.pv + ( int64_t( channel ) * p.F + t ) * ( C + 1 ) );
		const float2 * rowp = row + lane;
		#pragma unroll
		for( int q = 0; q < E; ++q ) mfr[q] = rowp[64 * q];
		mfny = row[C];
		};
	float2 mfr[E], mfny;
	load_row( t0, mfr, mfny );

	int64_t pos = chain_start;
	for( int64_t t = t0; t < t1; ++t )
		{
		// ---- inverse phase vocoder per bin (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61)
		float2 z[E];
		float mn;
		bool slow = false;
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const float2 mf = mfr[q];                                           // ( m, f )
			ph[q] += double( mf.y / p.analysis_rate * FLANHIP_PI2_F );          // phase_vocoder.cpp:57-58
			slow |= !( __builtin_fabs( ph[q] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
			z[q].x = mf.x;
			}
			{
			const float2 mf = mfny;
			ph[E] += double( mf.y / p.analysis_rate * FLANHIP_PI2_F );
			slow |= !( __builtin_fabs( ph[E] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
			mn = mf.x;
			}
		if( t + 1 < t1 ) load_row( t + 1, mfr, mfny );                          // prefetch: in flight during the transform below
		float2 xn;
		if( __any( slow ) )
			{
			// a phase outside the range the fast helpers are exact for (or a NaN): the general routines for this frame
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				ph[q] = fold_phase_any( ph[q] );
				const float2 sc = sincos_wide( float( ph[q] ) );
				z[q] = make_float2( z[q].x * sc.y, z[q].x * sc.x );
				}
			ph[E] = fold_phase_any( ph[E] );
			const float2 sc = sincos_wide( float( ph[E] ) );
			xn = make_float2( mn * sc.y, mn * sc.x );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				ph[q] = fold_phase_fast( ph[q] );                               // phase_vocoder.cpp:59
				float sn, cs;
				sincos_fast( float( ph[q] ), sn, cs );
				z[q] = make_float2( z[q].x * cs, z[q].x * sn );                 // std::polar, :60
				__builtin_amdgcn_sched_barrier( 0 );                            // one bin at a time: keeps the temporaries of 16 bins from overlapping
				}
			ph[E] = fold_phase_fast( ph[E] );
			float sn, cs;
			sincos_fast( float( ph[E] ), sn, cs );
			xn = make_float2( mn * cs, mn * sn );
			}
		// ---- merge X[0..C] into the half-size spectrum: needs X[k] (own) and X[C-k] (mirror lane): one LDS exchange
		#pragma unroll
		for( int q = 0; q < E; ++q ) buf[padl + 68 * q] = z[q];
		if( lane == 0 ) buf[( C * 17 ) / 16] = xn;                              // slot PAD( C )
		wave_sync();
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			float2 xk = z[q];
			float2 xm = mirror[-68 * q];                                        // X[ C - k ]; k = 0 pairs with X[C]
			if( q == 0 ) { xk.y = ( lane == 0 ) ? 0.0f : xk.y; xm.y = ( lane == 0 ) ? 0.0f : xm.y; }   // c2r ignores Im X[0], Im X[C]
			const float2 w = s_w2[64 * q];
			const float ax = xk.x + xm.x, ay = xk.y - xm.y;                     // A = X[k] + conj X[C-k]
			const float dx = xk.x - xm.x, dy = xk.y + xm.y;                     // D = X[k] - conj X[C-k]
			const float c = w.x, sgn = -w.y;                                    // exp(+2 pi i k / N)
			const float bx = __builtin_fmaf( c, dx, -( sgn * dy ) ), by = __builtin_fmaf( c, dy, sgn * dx );
			z[q] = make_float2( ax - by, -( ay + bx ) );                        // conj( A + iB ): forward FFT of it = conj of the inverse
			__builtin_amdgcn_sched_barrier( 0 );
			}
		wave_sync();
		fft_fast<LOG2C>( z, buf, s_tw1, s_tw3, lane );
		// ---- G = fft( conj Z ): x[2n] = G[n].x, x[2n+1] = -G[n].y; window and overlap-add (AudioPV.cpp:122-134)
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const float2 w = s_win[64 * q];                                     // zero beyond W
			acc[q].x += z[q].x * w.x;
			acc[q].y += ( -z[q].y ) * w.y;
			}
		#pragma unroll
		for( int q = 0; q < HOPQ; ++q ) emit_step( pos + 128 * q, acc[q] );
		#pragma unroll
		for( int q = 0; q < E; ++q ) acc[q] = ( q + HOPQ < E ) ? acc[q + HOPQ] : make_float2( 0.0f, 0.0f );
		pos += hop;
		}
	// flush the partial sums that the next chain's head completes; the last chain zero-fills to the end of the output
	const int64_t ring_end = pos + ( W - hop );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	#pragma unroll
	for( int q = 0; q < E; ++q )
		{
		const int64_t a0 = pos + 128 * q;
		if( a0 < flush_end ) emit_step( a0, acc[q] );
		}
	for( int64_t a0 = pos + 128 * E; a0 < flush_end; a0 += 128 ) emit_step( a0, make_float2( 0.0f, 0.0f ) );
	}

// ---- faster pre-pass kernels (all sizes) -----------------------------------------------------------------------
// One thread per (chain, bin): sums the chain's phase increments, folded like the running phase; NaN/Inf scan.
__global__ __launch_bounds__( 256 ) void k_phase_sums2( SynthParams p )
	{
	const int64_t chain = blockIdx.x;
	const int k = blockIdx.y * blockDim.x + threadIdx.x;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int n = int( min( t0 + int64_t( p.L ), p.F ) - t0 );
	bool bad = false;
	if( k < p.num_bins )
		{
		double ph = 0.0;
		const float2 * col = reinterpret_cast<const float2*>( p.pv + ( int64_t( channel ) * p.F + t0 ) * p.num_bins + k );
		int i = 0;
		for( ; i + 4 <= n; i += 4 )
			{
			float2 v[4];
			#pragma unroll
			for( int u = 0; u < 4; ++u ) v[u] = col[int64_t( i + u ) * p.num_bins];
			#pragma unroll
			for( int u = 0; u < 4; ++u )
				{
				bad |= isnan( v[u].x ) || isnan( v[u].y ) || isinf( v[u].x ) || isinf( v[u].y );
				ph += double( v[u].y / p.analysis_rate * FLANHIP_PI2_F );
				ph = ( __builtin_fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph );
				}
			}
		for( ; i < n; ++i )
			{
			const float2 v = col[int64_t( i ) * p.num_bins];
			bad |= isnan( v.x ) || isnan( v.y ) || isinf( v.x ) || isinf( v.y );
			ph += double( v.y / p.analysis_rate * FLANHIP_PI2_F );
			ph = ( __builtin_fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph );
			}
		p.carry[chain * p.num_bins + k] = ph;
		}
	if( p.nan_flag && __any( bad ) && ( threadIdx.x & 63 ) == 0 ) atomicOr( p.nan_flag, 1 );
	}

// Exclusive scan of the chain sums along each channel, per bin.
__global__ __launch_bounds__( 256 ) void k_phase_scan2( SynthParams p )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= int64_t( p.num_channels ) * p.num_bins ) return;
	const int channel = int( idx / p.num_bins ), k = int( idx % p.num_bins );
	double * c = p.carry + int64_t( channel ) * p.chains_per_channel * p.num_bins + k;
	double run = 0.0;                                                          // AudioPV.cpp:111
	const int n = p.chains_per_channel;
	int i = 0;
	for( ; i + 8 <= n; i += 8 )
		{
		double v[8];
		#pragma unroll
		for( int u = 0; u < 8; ++u ) v[u] = c[int64_t( i + u ) * p.num_bins];
		#pragma unroll
		for( int u = 0; u < 8; ++u )
			{
			c[int64_t( i + u ) * p.num_bins] = run;
			run += v[u];
			run = ( __builtin_fabs( run ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( run ) : fold_phase_any( run );
			}
		}
	for( ; i < n; ++i )
		{
		const double v = c[int64_t( i ) * p.num_bins];
		c[int64_t( i ) * p.num_bins] = run;
		run += v;
		run = ( __builtin_fabs( run ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( run ) : fold_phase_any( run );
		}
	}

} // namespace flanhip
