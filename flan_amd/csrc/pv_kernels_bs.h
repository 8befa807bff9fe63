// pv_kernels_bs.h -- Audio::convert_to_PV / PV::convert_to_audio for dft sizes with a large prime factor (2998 = 2 x 1499 ...), which FFTW plans
// in O( N log N ) like any other size (reference: FFTHelper.cpp:16-26) and which ran as direct sums here until round 5 (pv_kernels_any.h: 35 ms
// for 8 ch x 60 s at dft 2998).  bs_plan.h says which sizes and how.
//
// Structure: the mixed-radix kernels' (pv_kernels_mr.h: one block of 512 threads walks a chain of frames, the transform shared in LDS, what crosses
// frames kept in LDS) with the C-point complex transform replaced by Bluestein's chirp-z form through two power-of-two transforms of M >= 2 C - 1:
//     a[n] = z[n] conj( w[n] )  (n < C; zero up to M)        w[n] = exp( + pi i n^2 / C ), table `chirp`, evaluated in double from n^2 mod 2 C
//     A = fft_M( a );   P = A . Bh                           Bh = fft_M( b ) / M in double,  b[m] = b[M - m] = w[m] for m < C, zero between
//     R = fft_M( conj P )                                    ( = conj of the inverse transform: the convolution's conjugate )
//     Z[k] = conj( w[k] R[k] ),  k < C
// Nothing of this is a pass of its own: the chirp multiplies ride on the loop that fills the frame and on the loops that consume the result, the
// zero padding is the first pass of the first transform not reading beyond C, and A . Bh (conjugated) is how the first pass of the second transform
// reads its input.  All in fp32 like every transform here: 2 log2( M ) + 3 roundings per point instead of log2( C ) -- the bit-identical share of f
// the tests hold these sizes to is lower for that (tests/test_gpu_conversions.py), the magnitudes and the audio are not measurably further off.
#pragma once
#include "pv_kernels_mr.h"
#include "bs_plan.h"

namespace flanhip {

struct BsTables
	{
	const cf * tw;        // [M]  exp( -2 pi i j / M )
	const d2 * chirp;     // [C]  w[n], in double
	const d2 * bh;        // [M]  fft_M( b ) / M, in double
	unsigned char * scratch = nullptr;   // BsPlan::glob: [blocks][scratch_stride] bytes -- a block's two buffers and what it keeps across frames (bsg_block_bytes)
	size_t scratch_stride = 0;
	};
// BsPlan::glob, bytes of device memory per block: two frames of padded_len( M + 1 ) points, running phases / phase sums ( C + 2 doubles ), previous phases ( C + 4
// floats ), the synthesis' ring ( W floats )
inline size_t bsg_block_bytes( int C, int M, int W ) { return ( 2 * size_t( padded_len( M + 1 ) ) * 8 + size_t( C + 2 ) * 8 + size_t( C + 4 ) * 4 + size_t( ( W + 3 ) & ~3 ) * 4 + 255 ) & ~size_t( 255 ); }
// a float point times a table entry kept in double, rounded once: the three chirp products of a transform cost three roundings, not nine (in
// fp32, tables and products, the share of f bit for bit the oracle's fell just under the floor the tests hold sizes like 2998 to: 0.978 for 0.98)
__device__ __forceinline__ cf cmul_d( cf a, d2 w )
	{
	const double ax = double( a.x ), ay = double( a.y );
	return mk( float( __builtin_fma( ax, w.x, -( ay * w.y ) ) ), float( __builtin_fma( ax, w.y, ay * w.x ) ) );
	}
__device__ __forceinline__ d2 conj_d( d2 w ) { return d2{ w.x, -w.y }; }
__device__ __forceinline__ cf conj_f( cf a ) { return mk( a.x, -a.y ); }

inline bool bs_pingpong( int M ) { return M <= BS_PP_MAX_M; }
// LDS of one block (bytes): [twiddles M cf, ping-pong sizes][frame padded_len( M + 1 ) cf, two with ping-pong][state]
inline size_t bs_lds_common( int M ) { return ( bs_pingpong( M ) ? size_t( M ) * 8 : 0 ) + ( bs_pingpong( M ) ? 2 : 1 ) * size_t( padded_len( M + 1 ) ) * 8; }
inline size_t bs_analyze_lds( int C, int M ) { return bs_lds_common( M ) + ( bs_pingpong( M ) ? size_t( C + 1 ) * 8 : 0 ) + size_t( C + 4 ) * 4; }       // phase sums (ping-pong sizes), previous phases
inline size_t bs_synth_lds( int C, int M, int W, bool win_lds ) { return bs_lds_common( M ) + ( win_lds ? 2 : 1 ) * size_t( ( W + 3 ) & ~3 ) * 4 + size_t( C + 2 ) * 8; }   // ring (+ scaled window), running phases

// The first pass of a transform (radix 8, sub-transform length 1: no twiddles), reading its input through what Bluestein needs there:
// MUL = false: points from `limit` on are zero (never read);  MUL = true: conj( src[i] bh[i] ).
// (bhr: the thread's own eight entries of bh held in registers by the caller -- HOIST kernels, whose M / 8 butterflies are at most one per thread)
template<bool PP, bool MUL> __device__ __forceinline__ void bs_pass0( cf * src, cf * dst, int M, int limit, const d2 * __restrict__ bh, const d2 * bhr, int tid )
	{
	const int NB = M / 8;
	auto fetch = [&]( int idx, int r ) -> cf
		{
		if constexpr( MUL ) return conj_f( cmul_d( src[PAD( idx )], bhr ? bhr[r] : bh[idx] ) );
		else return idx < limit ? src[PAD( idx )] : mk( 0.0f, 0.0f );
		};
	if constexpr( PP )
		{
		#pragma unroll 1
		for( int j = tid; j < NB; j += MR_THREADS )
			{
			cf v[8];
			#pragma unroll
			for( int r = 0; r < 8; ++r ) v[r] = fetch( j + r * NB, r );
			dft_reg<8>( v );
			#pragma unroll
			for( int r = 0; r < 8; ++r ) dst[PAD( j * 8 + r )] = v[r];
			}
		__syncthreads();
		}
	else
		{
		constexpr int MAXB = MR_POINTS / 8;
		cf v[MAXB][8];
		#pragma unroll
		for( int b = 0; b < MAXB; ++b )
			{
			const int j = tid + MR_THREADS * b;
			if( j < NB )
				{
				#pragma unroll
				for( int r = 0; r < 8; ++r ) v[b][r] = fetch( j + r * NB, r );
				}
			}
		__syncthreads();
		#pragma unroll
		for( int b = 0; b < MAXB; ++b )
			{
			const int j = tid + MR_THREADS * b;
			if( j < NB )
				{
				dft_reg<8>( v[b] );
				#pragma unroll
				for( int r = 0; r < 8; ++r ) src[PAD( j * 8 + r )] = v[b][r];
				}
			}
		__syncthreads();
		}
	}

// ---- BsPlan::glob: the passes between two buffers in device memory.  Four butterflies per thread and trip, every load of the trip requested before the
// first butterfly is worked out (src and dst are different buffers: __restrict__ lets the loads pass the stores) -- a trip is one round trip to L2, not four
template<int R> __device__ __forceinline__ void bsg_pass( const cf * __restrict__ src, cf * __restrict__ dst, const cf * __restrict__ tw, int M, int NS, unsigned magic, int stride, int tid )
	{
	constexpr int B = R == 8 ? 4 : 8;
	const int NB = M / R;
	for( int j0 = tid; j0 < NB; j0 += B * MR_THREADS )
		{
		cf v[B][R];
		#pragma unroll
		for( int b = 0; b < B; ++b )
			{
			const int j = min( j0 + MR_THREADS * b, NB - 1 );
			#pragma unroll
			for( int r = 0; r < R; ++r ) v[b][r] = src[PAD( j + r * NB )];
			}
		#pragma unroll
		for( int b = 0; b < B; ++b )
			{
			const int j = j0 + MR_THREADS * b;
			if( j >= NB ) continue;
			const int k = NS > 1 ? j - int( __umulhi( unsigned( j ), magic ) ) * NS : 0;      // j % NS (exact up to M = 2^18: bs_plan.h)
			if( NS > 1 )
				{
				const int step = k * stride;
				#pragma unroll
				for( int r = 1; r < R; ++r ) v[b][r] = cmul( v[b][r], tw[r * step] );
				}
			dft_reg<R>( v[b] );
			const int base = ( j - k ) * R + k;
			#pragma unroll
			for( int r = 0; r < R; ++r ) dst[PAD( base + r * NS )] = v[b][r];
			}
		}
	__syncthreads();
	}
// the first pass (radix 8, no twiddles): MUL = false: points from `limit` on are zero; MUL = true: conj( src[i] bh[i] )
template<bool MUL> __device__ __forceinline__ void bsg_pass0( const cf * __restrict__ src, cf * __restrict__ dst, int M, int limit, const d2 * __restrict__ bh, int tid )
	{
	constexpr int B = 2;
	const int NB = M / 8;
	for( int j0 = tid; j0 < NB; j0 += B * MR_THREADS )
		{
		cf v[B][8];
		#pragma unroll
		for( int b = 0; b < B; ++b )
			{
			const int j = min( j0 + MR_THREADS * b, NB - 1 );
			#pragma unroll
			for( int r = 0; r < 8; ++r )
				{
				const int idx = j + r * NB;
				if constexpr( MUL ) v[b][r] = conj_f( cmul_d( src[PAD( idx )], bh[idx] ) );
				else v[b][r] = idx < limit ? src[PAD( idx )] : mk( 0.0f, 0.0f );
				}
			}
		#pragma unroll
		for( int b = 0; b < B; ++b )
			{
			const int j = j0 + MR_THREADS * b;
			if( j >= NB ) continue;
			dft_reg<8>( v[b] );
			#pragma unroll
			for( int r = 0; r < 8; ++r ) dst[PAD( j * 8 + r )] = v[b][r];
			}
		}
	__syncthreads();
	}
template<bool MUL> __device__ __forceinline__ cf * bsg_fft( cf * a, cf * b, const BsTables & tb, const BsPlan & pl, int limit, int tid )
	{
	cf * src = a, * dst = b;
	bsg_pass0<MUL>( src, dst, pl.M, limit, tb.bh, tid );
	{ cf * t = src; src = dst; dst = t; }
	int NS = 8;
	for( int i = 1; i < pl.npass; ++i )
		{
		const int r = pl.radix[i];
		switch( r )
			{
			case 8:  bsg_pass<8>( src, dst, tb.tw, pl.M, NS, pl.magic[i], pl.stride[i], tid ); break;
			case 4:  bsg_pass<4>( src, dst, tb.tw, pl.M, NS, pl.magic[i], pl.stride[i], tid ); break;
			default: bsg_pass<2>( src, dst, tb.tw, pl.M, NS, pl.magic[i], pl.stride[i], tid ); break;
			}
		NS *= r;
		{ cf * t = src; src = dst; dst = t; }
		}
	return src;
	}

// forward transform of the M points in `a` (natural order in and out), the first pass as above; returns where the result stands
template<bool PP, bool MUL> __device__ __forceinline__ cf * bs_fft( cf * a, cf * b, const BsTables & tb, const BsPlan & pl, int limit, int tid, const d2 * bhr = nullptr )
	{
	cf * src = a, * dst = PP ? b : a;
	bs_pass0<PP, MUL>( src, dst, pl.M, limit, tb.bh, bhr, tid );
	if constexpr( PP ) { cf * t = src; src = dst; dst = t; }
	int NS = 8;
	for( int i = 1; i < pl.npass; ++i )
		{
		const int r = pl.radix[i];
		switch( r )
			{
			case 8:  mr_pass<8, PP>( src, dst, tb.tw, pl.M, NS, pl.magic[i], pl.stride[i], tid ); break;
			case 4:  mr_pass<4, PP>( src, dst, tb.tw, pl.M, NS, pl.magic[i], pl.stride[i], tid ); break;
			default: mr_pass<2, PP>( src, dst, tb.tw, pl.M, NS, pl.magic[i], pl.stride[i], tid ); break;
			}
		NS *= r;
		if constexpr( PP ) { cf * t = src; src = dst; dst = t; }
		}
	return src;
	}

struct BsLds { BsTables tb; cf * buf, * buf2; unsigned char * state; };
template<bool PP, bool GLOB = false> __device__ __forceinline__ BsLds bs_carve( unsigned char * smem, const BsTables & g, int M, int tid )
	{
	BsLds l;
	if constexpr( GLOB )
		{
		// everything in the block's stretch of device memory; the twiddles stay where they are (read through L1 / L2)
		cf * base = reinterpret_cast<cf*>( g.scratch + size_t( blockIdx.x ) * g.scratch_stride );
		l.tb = g;
		l.buf = base;
		l.buf2 = base + padded_len( M + 1 );
		l.state = reinterpret_cast<unsigned char*>( base + 2 * padded_len( M + 1 ) );
		return l;
		}
	cf * s_tw = reinterpret_cast<cf*>( smem );
	l.tb = g;
	l.buf = s_tw + ( PP ? M : 0 );
	l.buf2 = l.buf + padded_len( M + 1 );
	l.state = reinterpret_cast<unsigned char*>( l.buf + ( PP ? 2 : 1 ) * padded_len( M + 1 ) );      // 8-byte aligned
	if constexpr( PP ) { for( int i = tid; i < M; i += MR_THREADS ) s_tw[i] = g.tw[i]; l.tb.tw = s_tw; }
	return l;
	}

// ---- Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78): one block per chain --------------------------------------------------------
// HOIST (ping-pong layout, C < 2048 = 4 x 512: every per-frame loop is ONE trip per thread): what a thread reads from the tables is the same every
// frame -- its eight entries of bh, its four chirp values, per-bin constants and window samples -- so it reads them once, into registers (72 of
// them: two wavefronts per SIMD, which is what one block per CU of the M = 4096 layout comes to anyway), and the next frame's samples are requested
// under this frame's bins.  Without it a frame waited for L2 three times (12.8 us a frame at dft 2998; the passes themselves are ~4).
// GLOB (round 6, BsPlan::glob): M = 16384 ... 2^18 -- the ping-pong kernel with its buffers and its state in device memory (no LDS at all)
template<bool PP, bool HOIST, bool GLOB = false>
__global__ __launch_bounds__( MR_THREADS, ( PP && !HOIST && !GLOB ) ? 4 : 2 ) void k_analyze_bs( AnalyzeParams p, BsPlan pl, BsTables g )
	{
	static_assert( PP || !HOIST, "HOIST is a ping-pong layout" );
	static_assert( !GLOB || ( PP && !HOIST ), "GLOB is a ping-pong layout" );
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int C = pl.C, M = pl.M, W = p.window_size, hop = p.hop, dft = 2 * C;
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	const BsLds l = bs_carve<PP, GLOB>( smem, g, M, tid );
	cf * buf = l.buf;
	const bool sums = PP && p.sums != nullptr;                                        // (k_analyze_mr: the fused round trip's pre-pass inside the kernel)
	double * s_sum = reinterpret_cast<double*>( l.state );                            // [C + 1] (PP)
	float * s_prev = reinterpret_cast<float*>( s_sum + ( PP ? C + 1 : 0 ) );          // [C + 1]
	bool bad = false;
	for( int k = tid; k <= C; k += MR_THREADS )
		{
		s_prev[k] = 0.0f;                                                              // AudioPV.cpp:44
		if( PP ) s_sum[k] = 0.0;
		}
	__syncthreads();
	if( s_cancel ) return;

	const int64_t chain = blockIdx.x;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                        // phase_vocoder.cpp:37
	constexpr int U = 4;
	cf * in = buf;                                                                     // (two transforms of the same number of passes: the result lands where the frame went in)
	// HOIST: the thread's constants, and the samples of the frame about to be transformed
	d2 chr[U], bhr[8]; v4f_t kcr[U]; float w0r[U], w1r[U], xa0[U], xa1[U];
	auto request = [&]( int64_t t )
		{
		const int64_t start = int64_t( hop ) * t - W / 2;
		#pragma unroll
		for( int u = 0; u < U; ++u )
			{
			const int i = tid + MR_THREADS * u;
			const int64_t p0 = start + 2 * i, p1 = p0 + 1;
			xa0[u] = ( i < C && 2 * i < W && p0 >= 0 && p0 < p.n ) ? x[p0] : 0.0f;
			xa1[u] = ( i < C && 2 * i + 1 < W && p1 >= 0 && p1 < p.n ) ? x[p1] : 0.0f;
			}
		};
	if constexpr( HOIST )
		{
		#pragma unroll
		for( int u = 0; u < U; ++u )
			{
			const int i = tid + MR_THREADS * u;
			chr[u] = g.chirp[min( i, C - 1 )];
			kcr[u] = mr_kc_value( min( i, C ), p.tw2[min( i, C )], p.sample_rate, p.analysis_rate, dft );
			w0r[u] = ( i < C && 2 * i < W ) ? p.window[2 * i] : 0.0f;
			w1r[u] = ( i < C && 2 * i + 1 < W ) ? p.window[2 * i + 1] : 0.0f;
			}
		#pragma unroll
		for( int r = 0; r < 8; ++r ) bhr[r] = g.bh[min( tid, M / 8 - 1 ) + r * ( M / 8 )];
		request( t0 > 0 ? t0 - 1 : t0 );
		}

	for( int64_t t = ( t0 > 0 ? t0 - 1 : t0 ); t < t1; ++t )                           // (the frame before the chain only lends its phases: phase_vocoder.cpp:45)
		{
		const bool emit = t >= t0;
		// a[n] = ( x[2n] win[2n], x[2n+1] win[2n+1] ) conj( w[n] ), n < C; zero beyond the window and outside the signal (AudioPV.cpp:52-65)
		const int64_t start = int64_t( hop ) * t - W / 2;
		cf * const other = PP ? ( in == buf ? l.buf2 : buf ) : buf;
		if constexpr( HOIST )
			{
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int i = tid + MR_THREADS * u;
				if( i < C ) in[PAD( i )] = cmul_d( mk( xa0[u] * w0r[u], xa1[u] * w1r[u] ), conj_d( chr[u] ) );   // AudioPV.cpp:60 (0 outside), then the chirp
				}
			}
		else for( int i0 = tid; i0 < C; i0 += U * MR_THREADS )
			{
			float a0[U], a1[U], w0[U], w1[U]; d2 ch[U];
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int i = i0 + MR_THREADS * u, s0 = 2 * i, s1 = 2 * i + 1;
				const int64_t p0 = start + s0, p1 = start + s1;
				const bool ok0 = i < C && s0 < W && p0 >= 0 && p0 < p.n, ok1 = i < C && s1 < W && p1 >= 0 && p1 < p.n;
				a0[u] = ok0 ? x[p0] : 0.0f; w0[u] = ok0 ? p.window[s0] : 0.0f;
				a1[u] = ok1 ? x[p1] : 0.0f; w1[u] = ok1 ? p.window[s1] : 0.0f;
				ch[u] = g.chirp[min( i, C - 1 )];
				}
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int i = i0 + MR_THREADS * u;
				if( i < C ) in[PAD( i )] = cmul_d( mk( a0[u] * w0[u], a1[u] * w1[u] ), conj_d( ch[u] ) );   // AudioPV.cpp:60, then the chirp
				}
			}
		__syncthreads();
		cf * const A = GLOB ? bsg_fft<false>( in, other, l.tb, pl, C, tid ) : bs_fft<PP, false>( in, other, l.tb, pl, C, tid );
		const cf * const R = GLOB ? bsg_fft<true>( A, A == in ? other : in, l.tb, pl, M, tid ) : bs_fft<PP, true>( A, A == in ? other : in, l.tb, pl, M, tid, HOIST ? bhr : nullptr );
		if constexpr( HOIST ) { if( t + 1 < t1 ) request( t + 1 ); }

		// the real transform's bins from the half-size one, each phase-vocoded (AudioPV.cpp:69-73);  Z[k] = conj( w[k] R[k] ),  w[C - k] = +- w[k]
		MF * row = p.out + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		const cf r0 = R[PAD( 0 )];
		const cf z0 = mk( r0.x, -r0.y );
		for( int k0 = tid; k0 <= C; k0 += U * MR_THREADS )
			{
			v4f_t kc[U]; d2 ch[U];
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				if constexpr( HOIST ) { kc[u] = kcr[u]; ch[u] = chr[u]; continue; }      // (one trip: k0 = tid)
				const int k = min( k0 + MR_THREADS * u, C );
				kc[u] = mr_kc_value( k, p.tw2[k], p.sample_rate, p.analysis_rate, dft );
				ch[u] = g.chirp[min( k, C - 1 )];
				}
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int k = k0 + MR_THREADS * u;
				if( k > C ) continue;
				float re, im;
				if( k == 0 ) { re = z0.x + z0.y; im = 0.0f; }
				else if( k == C ) { re = z0.x - z0.y; im = 0.0f; }
				else
					{
					// (the last chirp product and the split in double, rounded once: these roundings land on the bin itself)
					const double sg = double( pl.sign_c );
					const cf rk = R[PAD( k )], rm = R[PAD( C - k )];
					const double zkx = __builtin_fma( double( rk.x ), ch[u].x, -( double( rk.y ) * ch[u].y ) ), zky = -__builtin_fma( double( rk.x ), ch[u].y, double( rk.y ) * ch[u].x );
					const double zmx = sg * __builtin_fma( double( rm.x ), ch[u].x, -( double( rm.y ) * ch[u].y ) ), zmy = -sg * __builtin_fma( double( rm.x ), ch[u].y, double( rm.y ) * ch[u].x );
					const double ax = 0.5 * ( zkx + zmx ), ay = 0.5 * ( zky - zmy );
					const double dx = zkx - zmx, dy = zky + zmy;
					re = float( ax + 0.5 * __builtin_fma( double( kc[u].x ), dy, double( kc[u].y ) * dx ) );
					im = float( ay - 0.5 * __builtin_fma( double( kc[u].x ), dx, -( double( kc[u].y ) * dy ) ) );
					}
				float pr = s_prev[k];
				const MF mf = phase_vocode_bin( re, im, pr, kc[u].z, kc[u].w, p.analysis_rate, use_wrapping );
				s_prev[k] = pr;
				if( emit )
					{
					row[k] = mf;
					if( PP && sums )
						{
						// k_phase_sums2 (pv_kernels_fast.h): the same terms in the same order, the partial sum folded where that kernel folds it
						double ph = s_sum[k] + double( div_c( mf.f, p.ar_div ) * FLANHIP_PI2_F );
						if( ( ( t - t0 ) & 7 ) == 7 && !( __builtin_fabs( ph ) < 1.0e8 ) ) ph = fold_phase_any( ph );
						s_sum[k] = ph;
						bad |= !( __builtin_fabsf( mf.m ) <= 3.4028235e38f ) || !( __builtin_fabsf( mf.f ) <= 3.4028235e38f );
						}
					}
				}
			}
		if constexpr( PP ) in = ( R == buf ) ? l.buf2 : buf;                            // the next frame goes where this spectrum is not (k_analyze_mr)
		else __syncthreads();
		}
	if( PP && sums )
		{
		double * dst = p.sums + chain * ( C + 1 );
		for( int k = tid; k <= C; k += MR_THREADS ) { const double ph = s_sum[k]; dst[k] = ( __builtin_fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph ); }
		const bool any_bad = __any( bad );
		if( p.nan_out && ( tid & 63 ) == 0 )
			{
			if( chain == 0 && tid == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
		}
	}

// ---- PV::convert_to_audio (Conversions/AudioPV.cpp:86-139): one block per chain, from the carries of the common pre-pass ------------------
// (HOIST as in k_analyze_bs: bh, chirp, split twiddles and the scaled window in registers; the next frame's PV row requested under the passes)
template<bool PP, bool HOIST, bool GLOB = false>
__global__ __launch_bounds__( MR_THREADS, ( PP && !HOIST && !GLOB ) ? 4 : 2 ) void k_synthesize_bs( SynthParams p, BsPlan pl, BsTables g )
	{
	static_assert( PP || !HOIST, "HOIST is a ping-pong layout" );
	static_assert( !GLOB || ( PP && !HOIST ), "GLOB is a ping-pong layout" );
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int C = pl.C, M = pl.M, W = p.window_size, hop = p.hop;
	const int wpad = ( W + 3 ) & ~3;
	const bool win_lds = pl.win_lds != 0;
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	const BsLds l = bs_carve<PP, GLOB>( smem, g, M, tid );
	cf * buf = l.buf;
	double * s_ph = reinterpret_cast<double*>( l.state );                             // [C + 2]
	float * ring = reinterpret_cast<float*>( s_ph + ( C + 2 ) );                       // [wpad]
	float * s_win = ring + wpad;                                                      // [wpad], if win_lds: the scaled window (AudioPV.cpp:102)
	auto win = [&]( int i ) { float v; if( win_lds ) { v = s_win[i]; asm volatile( "" : "+v"( v ) ); } else v = p.window[i] * p.window_scale; return v; };
	for( int i = tid; i < W; i += MR_THREADS ) { ring[i] = 0.0f; if( win_lds ) s_win[i] = p.window[i] * p.window_scale; }

	const int64_t chain = blockIdx.x;
	const int channel = int( chain / p.chains_per_channel );
	const int chain_in_channel = int( chain % p.chains_per_channel );
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	float * out = p.out + int64_t( channel ) * p.out_len;
	float * head = p.head + chain * p.head_len;
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;
	for( int k = tid; k <= C; k += MR_THREADS ) s_ph[k] = p.carry[chain * ( C + 1 ) + k];
	__syncthreads();
	if( s_cancel ) return;
	constexpr int U = 4;
	d2 chr[U], bhr[8]; cf war[2], wbr[2]; float wn0[U], wn1[U]; MF mfr[U];
	auto request = [&]( int64_t t )
		{
		const MF * row = p.pv + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		#pragma unroll
		for( int u = 0; u < U; ++u ) mfr[u] = row[min( tid + MR_THREADS * u, C )];
		};
	if constexpr( HOIST )
		{
		#pragma unroll
		for( int u = 0; u < U; ++u )
			{
			const int n = tid + MR_THREADS * u;
			chr[u] = g.chirp[min( n, C - 1 )];
			wn0[u] = 2 * n < W ? p.window[2 * n] * p.window_scale : 0.0f;                // AudioPV.cpp:102
			wn1[u] = 2 * n + 1 < W ? p.window[2 * n + 1] * p.window_scale : 0.0f;
			}
		#pragma unroll
		for( int u = 0; u < 2; ++u ) { const int k = min( tid + MR_THREADS * u, C / 2 ); war[u] = p.tw2[k]; wbr[u] = p.tw2[C - k]; }
		#pragma unroll
		for( int r = 0; r < 8; ++r ) bhr[r] = g.bh[min( tid, M / 8 - 1 ) + r * ( M / 8 )];
		request( t0 );
		}

	int ring_base = 0;
	int64_t pos = chain_start;
	for( int64_t t = t0; t < t1; ++t )
		{
		// inverse phase vocoder per bin (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61) -> X[0 .. C] in LDS
		const MF * row = p.pv + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		for( int k0 = tid; k0 <= C; k0 += U * MR_THREADS )
			{
			MF mfs[U];
			#pragma unroll
			for( int u = 0; u < U; ++u ) { if constexpr( HOIST ) mfs[u] = mfr[u]; else mfs[u] = row[min( k0 + MR_THREADS * u, C )]; }   // (HOIST: one trip, k0 = tid)
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int k = k0 + MR_THREADS * u;
				if( k > C ) continue;
				const MF mf = mfs[u];
				const double term = double( div_c( mf.f, p.ar_div ) * FLANHIP_PI2_F );    // :57-58
				double phase = s_ph[k] + term;
				phase = ( __builtin_fabs( phase ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( phase ) : fold_phase_any( phase );   // :59
				s_ph[k] = phase;
				const float th = float( phase );
				float sn, cs;
				if( __builtin_fabsf( th ) < FLANHIP_SINCOS_FAST_LIMIT ) sincos_fast( th, sn, cs );
				else { const float2 sc = sincos_wide( th ); sn = sc.x; cs = sc.y; }
				buf[PAD( k )] = mk( mf.m * cs, mf.m * sn );                               // std::polar, :60
				}
			}
		if constexpr( HOIST ) { if( t + 1 < t1 ) request( t + 1 ); }
		__syncthreads();
		// merge X[0 .. C] into the conjugated half-size spectrum (k_synthesize_mr), each point times conj( w[k] ) on the way: a[k] of the chirp-z form
		auto merge = [&]( int k, cf xk, cf xm, cf w2q ) -> cf
			{
			if( k == 0 ) { xk.y = 0.0f; xm.y = 0.0f; }
			const float ax = xk.x + xm.x, ay = xk.y - xm.y;
			const float dx = xk.x - xm.x, dy = xk.y + xm.y;
			const float c = w2q.x, s = -w2q.y;
			const float bx = __builtin_fmaf( c, dx, -( s * dy ) ), by = __builtin_fmaf( c, dy, s * dx );
			return mk( ax - by, -( ay + bx ) );
			};
		for( int k0 = tid; 2 * k0 <= C; k0 += U * MR_THREADS )
			{
			cf wa[U], wb[U]; d2 ch[U];
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				if constexpr( HOIST ) { wa[u] = war[u & 1]; wb[u] = wbr[u & 1]; ch[u] = chr[u]; continue; }   // (one trip; 2 k <= C < 2048: u < 2 is all there is)
				const int k = min( k0 + MR_THREADS * u, C / 2 ); wa[u] = p.tw2[k]; wb[u] = p.tw2[C - k]; ch[u] = g.chirp[k];
				}
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int k = k0 + MR_THREADS * u;
				if( 2 * k > C ) continue;
				const cf xk = buf[PAD( k )], xm = buf[PAD( C - k )];
				const d2 cw = conj_d( ch[u] );                                              // conj( w[k] );  conj( w[C - k] ) = +- the same
				const double sg = double( pl.sign_c );
				const cf zk = merge( k, xk, xm, wa[u] );
				if( k != 0 && 2 * k != C ) buf[PAD( C - k )] = cmul_d( merge( C - k, xm, xk, wb[u] ), d2{ cw.x * sg, cw.y * sg } );
				buf[PAD( k )] = cmul_d( zk, cw );
				}
			}
		__syncthreads();
		cf * const A = GLOB ? bsg_fft<false>( buf, l.buf2, l.tb, pl, C, tid ) : bs_fft<PP, false>( buf, l.buf2, l.tb, pl, C, tid );
		const cf * const R = GLOB ? bsg_fft<true>( A, A == buf ? l.buf2 : buf, l.tb, pl, M, tid ) : bs_fft<PP, true>( A, A == buf ? l.buf2 : buf, l.tb, pl, M, tid, HOIST ? bhr : nullptr );   // (lands in buf: two transforms of equally many passes)
		// G[n] = conj( w[n] R[n] ) = fft_C( conj Z )[n]:  x[2n] = G[n].x, x[2n+1] = -G[n].y (AudioPV.cpp:122); window, accumulate (:133-134)
		auto accumulate = [&]( int n, d2 ch, float wa0, float wa1 )
			{
			const cf gn = conj_f( cmul_d( R[PAD( n )], ch ) );
			int i0 = ring_base + 2 * n; if( i0 >= W ) i0 -= W;
			ring[i0] += gn.x * wa0;
			if( 2 * n + 1 < W )
				{
				int i1 = i0 + 1; if( i1 >= W ) i1 -= W;
				ring[i1] += ( -gn.y ) * wa1;
				}
			};
		if constexpr( HOIST )
			{
			#pragma unroll
			for( int u = 0; u < U; ++u ) { const int n = tid + MR_THREADS * u; if( 2 * n < W ) accumulate( n, chr[u], wn0[u], wn1[u] ); }   // ( W <= 2 C < 4096 )
			}
		else for( int n = tid; 2 * n < W; n += MR_THREADS ) accumulate( n, g.chirp[n], win( 2 * n ), 2 * n + 1 < W ? win( 2 * n + 1 ) : 0.0f );
		__syncthreads();
		for( int e = tid; e < hop; e += MR_THREADS )
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
		pos += hop;
		ring_base = ( hop < W ) ? ring_base + hop : 0;
		if( ring_base >= W ) ring_base -= W;
		}
	const int64_t ring_end = pos + ( hop < W ? W - hop : 0 );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	for( int64_t a = pos + tid; a < flush_end; a += MR_THREADS )
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

} // namespace flanhip
