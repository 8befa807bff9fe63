// pv_kernels_mr.h -- Audio::convert_to_PV / PV::convert_to_audio for the dft sizes FFTW would plan in O( N log N ) and the power-of-two kernels
// do not serve: every even size whose half C = N / 2 factors into 2, 3, 5, 7, 11 and 13 with C <= 8192 -- 3000, 1000, 66, 6, 1920, 6000 ...
// and the power of two 16384 (reference: FFTHelper.cpp:16-26 hands any size to fftwf_plan_dft_r2c_1d / c2r_1d; Audio.h:158-163 constrains
// nothing).  Round 3 ran all of them as direct sums (pv_kernels_any.h: O( window x bins ) per frame, 35 ms for 8 ch x 60 s at dft 3000);
// those stay for sizes with a larger prime factor (2998 = 2 x 1499) and beyond 16384.
//
// Structure: the generic kernels' (pv_kernels.h) with everything that was a template constant there a run-time value here.  A chain of frames
// is walked by one block of 512 threads sharing one transform in LDS: the real frame as C complex points, a Stockham pass per radix of the
// plan, then the real-transform split and the per-bin phase-vocoder arithmetic of pv_kernels.h, rounding for rounding.  Up to C = 4096 a
// pass goes from one LDS buffer to a second one (PP: a thread works its butterflies off one after the other -- gather, twiddles, a register
// DFT of the radix, scatter -- 70 registers, four blocks per CU, one barrier per pass); beyond, where two buffers do not fit, in place (every
// thread reads ALL its butterflies' inputs into registers -- up to 16 points --, barrier, scatter, barrier: 256 registers, one block per CU).
// What crosses frames (previous phases; running phases and the overlap-add ring in the synthesis) lives in LDS, the per-bin loops are rolled.
// Twiddles exp( -2 pi i j / C ) come from LDS while C <= 4096 and through L1 / L2 beyond (a 64 KB table does not fit beside a 64 KB frame).
#pragma once
#include "pv_kernels.h"
#include "mr_consts.h"

namespace flanhip {

constexpr int MR_THREADS = 512, MR_MAX_C = 8192, MR_MAX_PASSES = 14, MR_POINTS = MR_MAX_C / MR_THREADS;   // points per thread a pass may hold
constexpr int MR_TW_LDS_MAX_C = 4096;

struct MrPlan
	{
	int C, npass;
	int win_lds;                                 // synthesis: the scaled window in LDS (not for C = 8192 with a 4096 window: read through L1 there)
	int kc_lds;                                  // per-bin constants { split twiddle, bin frequency, expected phase advance } in LDS (where that costs no resident block)
	int ring_ws;                                 // synthesis: the overlap-add ring in the workspace (SynthParams::ring_ws), for windows it does not fit the LDS with (round 6:
	                                             // ( 8192, 512, 16384 ) off the team grid and every window above ~6000 at dft 16384 ran the direct sums, 0.8 s for 8 ch x 60 s)
	unsigned char radix[MR_MAX_PASSES];          // C = product; 16s first (in-place plans only; their passes want the short sub-transform lengths), then 8 / 4 / 2, then the odd ones
	unsigned magic[MR_MAX_PASSES];               // floor( 2^32 / NS ) + 1 of the pass (NS = the product of the radices before it): j / NS = mulhi( j, magic ) for j < 2^13
	unsigned short stride[MR_MAX_PASSES];        // C / ( NS R ) of the pass: the twiddle step per position inside a sub-transform (no division in the kernel)
	};

// C = 2^a 3^b 5^c 7^d 11^e 13^f <= 8192 ?  (host and device agree through the struct)
inline bool mr_make_plan( int dft, MrPlan * out )
	{
	if( dft < 4 || dft % 2 ) return false;
	int C = dft / 2;
	if( C > MR_MAX_C ) return false;
	MrPlan pl{};
	pl.C = C;
	int rest = C, n = 0;
	auto take = [&]( int r ) { while( rest % r == 0 && n < MR_MAX_PASSES ) { pl.radix[n++] = (unsigned char) r; rest /= r; } };
	if( C > MR_TW_LDS_MAX_C ) take( 16 );            // (ping-pong passes -- C <= 4096 -- run at 128 registers: radix 8 at most, see mr_fft)
	take( 8 ); take( 4 ); take( 2 ); take( 13 ); take( 11 ); take( 7 ); take( 5 ); take( 3 );
	if( rest != 1 ) return false;
	pl.npass = n;
	for( int i = 0, NS = 1; i < n; NS *= pl.radix[i], ++i )
		{
		pl.magic[i] = NS > 1 ? unsigned( ( uint64_t( 1 ) << 32 ) / unsigned( NS ) ) + 1u : 0u;
		pl.stride[i] = (unsigned short) ( C / ( NS * pl.radix[i] ) );
		}
	*out = pl;
	return true;
	}

// LDS of one block (bytes): [twiddles C cf if C <= 4096][window W floats: synthesis, if win_lds][frame padded_len( C + 1 ) cf, two with PP]
// [per-bin constants ( C + 1 ) float4 if kc_lds][state]
inline bool mr_pingpong( int C ) { return C <= MR_TW_LDS_MAX_C; }
inline size_t mr_lds_common( int C, bool kc_lds ) { return ( C <= MR_TW_LDS_MAX_C ? size_t( C ) * 8 : 0 ) + ( mr_pingpong( C ) ? 2 : 1 ) * size_t( padded_len( C + 1 ) ) * 8 + ( kc_lds ? size_t( C + 1 ) * 16 : 0 ); }
inline size_t mr_analyze_lds( int C, int W, bool win_lds, bool kc_lds ) { (void) W; (void) win_lds; return mr_lds_common( C, kc_lds ) + ( mr_pingpong( C ) ? size_t( C + 1 ) * 8 : 0 ) + size_t( C + 4 ) * 4; }   // the chain's phase sums (ping-pong sizes: the fused round trip's pre-pass, kept by the kernel itself), previous phases (the window is read from memory beside the samples)
inline size_t mr_synth_lds( int C, int W, bool win_lds, bool kc_lds, bool ring_ws = false ) { return mr_lds_common( C, kc_lds ) + ( ( win_lds ? 1 : 0 ) + ( ring_ws ? 0 : 1 ) ) * size_t( ( W + 3 ) & ~3 ) * 4 + size_t( C + 2 ) * 8; }   // scaled window, ring (unless in the workspace), running phases

// ---- register DFTs: powers of two from fft_device.h, odd primes by the symmetric direct form ---------------------------------------
// X[k] = v0 + sum_j ( a_j cos( 2 pi j k / R ) - i b_j sin( 2 pi j k / R ) ),  a_j = v[j] + v[R-j],  b_j = v[j] - v[R-j],  j = 1 .. (R-1)/2;
// X[R-k] is the same with + i.  Indices are compile-time after unrolling: the constants fold into the instructions.
template<int R> __device__ __forceinline__ void dft_odd( cf * v )
	{
	constexpr int H = ( R - 1 ) / 2;
	cf a[H], b[H];
	#pragma unroll
	for( int j = 1; j <= H; ++j ) { a[j - 1] = cadd( v[j], v[R - j] ); b[j - 1] = csub( v[j], v[R - j] ); }
	cf sum = v[0];
	#pragma unroll
	for( int j = 0; j < H; ++j ) sum = cadd( sum, a[j] );
	cf out[R];
	out[0] = sum;
	#pragma unroll
	for( int k = 1; k <= H; ++k )
		{
		cf p = v[0], q = mk( 0.0f, 0.0f );
		#pragma unroll
		for( int j = 1; j <= H; ++j )
			{
			const float c = OddRadix<R>::c[( j * k ) % R], s = OddRadix<R>::s[( j * k ) % R];
			p = mk( __builtin_fmaf( a[j - 1].x, c, p.x ), __builtin_fmaf( a[j - 1].y, c, p.y ) );
			q = mk( __builtin_fmaf( b[j - 1].x, s, q.x ), __builtin_fmaf( b[j - 1].y, s, q.y ) );
			}
		// -i q = ( q.y, -q.x )
		out[k] = mk( p.x + q.y, p.y - q.x );
		out[R - k] = mk( p.x - q.y, p.y + q.x );
		}
	#pragma unroll
	for( int k = 0; k < R; ++k ) v[k] = out[k];
	}
template<int R> __device__ __forceinline__ void dft_any_radix( cf * v )
	{
	if constexpr( R == 2 || R == 4 || R == 8 || R == 16 ) dft_reg<R>( v ); else dft_odd<R>( v );
	}

// One Stockham pass of radix R over C points, sub-transform length NS on entry (fft_device.h: fft_pass, with C and NS run-time).
// PP: from `src` to `dst`, butterfly by butterfly; otherwise in place in `src`.
template<int R, bool PP> __device__ __forceinline__ void mr_pass( cf * src, cf * dst, const cf * __restrict__ tw, int C, int NS, unsigned magic, int stride, int tid )
	{
	auto mod_ns = [&]( int j ) { return j - int( __umulhi( unsigned( j ), magic ) ) * NS; };   // j % NS, j < 2^13 (MrPlan::magic)
	const int NB = C / R;
	if constexpr( PP )
		{
		#pragma unroll 1
		for( int j = tid; j < NB; j += MR_THREADS )
			{
			cf v[R];
			#pragma unroll
			for( int r = 0; r < R; ++r ) v[r] = src[PAD( j + r * NB )];
			const int k = NS > 1 ? mod_ns( j ) : 0;
			if( NS > 1 )
				{
				const int step = k * stride;
				#pragma unroll
				for( int r = 1; r < R; ++r ) v[r] = cmul( v[r], tw[r * step] );           // r k stride < C
				}
			dft_any_radix<R>( v );
			const int base = ( j - k ) * R + k;
			#pragma unroll
			for( int r = 0; r < R; ++r ) dst[PAD( base + r * NS )] = v[r];
			}
		__syncthreads();
		}
	else
		{
		constexpr int MAXB = ( MR_POINTS + R - 1 ) / R;                              // butterflies a thread may own
		cf v[MAXB][R];
		#pragma unroll
		for( int b = 0; b < MAXB; ++b )
			{
			const int j = tid + MR_THREADS * b;
			if( j < NB )
				{
				#pragma unroll
				for( int r = 0; r < R; ++r ) v[b][r] = src[PAD( j + r * NB )];
				}
			}
		__syncthreads();
		#pragma unroll
		for( int b = 0; b < MAXB; ++b )
			{
			const int j = tid + MR_THREADS * b;
			if( j < NB )
				{
				const int k = NS > 1 ? mod_ns( j ) : 0;
				if( NS > 1 )
					{
					const int step = k * stride;
					#pragma unroll
					for( int r = 1; r < R; ++r ) v[b][r] = cmul( v[b][r], tw[r * step] );
					}
				dft_any_radix<R>( v[b] );
				const int base = ( j - k ) * R + k;
				#pragma unroll
				for( int r = 0; r < R; ++r ) src[PAD( base + r * NS )] = v[b][r];
				}
			}
		__syncthreads();
		}
	}

// forward complex transform of the C points in `a` (padded layout, natural order in and out); every thread of the block calls it.  Returns where
// the result stands: `a`, or with PP and an odd number of passes the second buffer `b`
// Which radices a kernel is COMPILED with decides its registers: with the radix-16 butterfly or the 7- / 11- / 13-point forms in the switch the
// 128-register ping-pong kernels spilled 26 / 38 registers -- loop invariants, reloaded one `scratch_load` + `s_waitcnt vmcnt(0)` at a time at the head
// of EVERY pass (ten memory round trips per pass: dft 3000 spent half its time there).  So: ping-pong plans hold no 16 (mr_make_plan), and the large
// odd radices are compiled into a second instantiation (BIG) that only plans containing one of them take.
template<bool PP, bool BIG> __device__ __forceinline__ cf * mr_fft( cf * a, cf * b, const cf * __restrict__ tw, const MrPlan & pl, int tid )
	{
	int NS = 1;
	cf * src = a, * dst = PP ? b : a;
	for( int i = 0; i < pl.npass; ++i )
		{
		const int r = pl.radix[i];
		bool done = true;
		switch( r )
			{
			case 8:  mr_pass<8, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
			case 4:  mr_pass<4, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
			case 2:  mr_pass<2, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
			case 3:  mr_pass<3, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
			case 5:  mr_pass<5, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
			default: done = false; break;
			}
		if constexpr( !PP ) { if( !done && r == 16 ) { mr_pass<16, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); done = true; } }
		if constexpr( BIG )
			{
			if( !done ) switch( r )
				{
				case 7:  mr_pass<7, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
				case 11: mr_pass<11, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
				default: mr_pass<13, PP>( src, dst, tw, pl.C, NS, pl.magic[i], pl.stride[i], tid ); break;
				}
			}
		NS *= r;
		if constexpr( PP ) { cf * t = src; src = dst; dst = t; }
		}
	return src;
	}
inline bool mr_plan_is_big( const MrPlan & pl ) { for( int i = 0; i < pl.npass; ++i ) if( pl.radix[i] == 7 || pl.radix[i] == 11 || pl.radix[i] == 13 ) return true; return false; }
// (what mr_fft's instantiations rely on: a ping-pong plan holds no radix 16; checked where a kernel is chosen)
inline bool mr_plan_fits_kernels( const MrPlan & pl ) { if( !mr_pingpong( pl.C ) ) return true; for( int i = 0; i < pl.npass; ++i ) if( pl.radix[i] == 16 ) return false; return true; }

// LDS carve-up shared by the two kernels
struct MrLds { const cf * tw; cf * buf, * buf2; v4f_t * kc; unsigned char * state; };
template<bool PP> __device__ __forceinline__ MrLds mr_carve( unsigned char * smem, const cf * g_tw, int C, bool kc_lds, int tid )
	{
	MrLds l;
	cf * s_tw = reinterpret_cast<cf*>( smem );
	constexpr bool tw_lds = PP;                                                     // ( C <= MR_TW_LDS_MAX_C exactly when the passes ping-pong: mr_pingpong ) -- a compile-time fact, so
	                                                                                // that the passes' twiddle reads are LDS instructions, not flat ones waited for one by one
	l.buf = s_tw + ( tw_lds ? C : 0 );
	l.buf2 = l.buf + padded_len( C + 1 );                                            // (PP: the passes' second buffer)
	l.kc = reinterpret_cast<v4f_t*>( l.buf + ( PP ? 2 : 1 ) * padded_len( C + 1 ) );   // (padded_len( C + 1 ) cf from a 16-byte aligned start: still 8-byte aligned -- see mr_kc)
	l.state = reinterpret_cast<unsigned char*>( reinterpret_cast<cf*>( l.kc ) + ( kc_lds ? 2 * ( C + 1 ) : 0 ) );
	if constexpr( tw_lds ) { for( int i = tid; i < C; i += MR_THREADS ) s_tw[i] = g_tw[i]; l.tw = s_tw; }
	else l.tw = g_tw;
	return l;
	}
// per-bin constants of bin k: { split twiddle exp( -pi i k / C ), bin frequency (PVBuffer.cpp:443-446), expected phase advance (phase_vocoder.cpp:47) }
// -- from the LDS table (two 8-byte reads: the table is only 8-byte aligned) or worked out on the spot
__device__ __forceinline__ v4f_t mr_kc_value( int k, cf w2, float sample_rate, float analysis_rate, int dft )
	{
	const float binf = float( k ) * sample_rate / float( dft );
	return v4f_t{ w2.x, w2.y, binf, binf / analysis_rate * FLANHIP_PI2_F };
	}
__device__ __forceinline__ v4f_t mr_kc( const MrLds & l, bool kc_lds, const cf * __restrict__ g_tw2, int k, float sample_rate, float analysis_rate, int dft )
	{
	if( kc_lds )
		{
		const cf * t = reinterpret_cast<const cf*>( l.kc ) + 2 * k;
		const cf a = t[0], b = t[1];
		return v4f_t{ a.x, a.y, b.x, b.y };
		}
	return mr_kc_value( k, g_tw2[k], sample_rate, analysis_rate, dft );
	}

// ---- Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78): one block per chain --------------------------------------------------------
// (PP: four wavefronts per SIMD = 128 registers, the rolled passes need ~70; in place: two per SIMD = 256 registers for the passes' unrolled butterflies)
template<bool PP, bool BIG>
__global__ __launch_bounds__( MR_THREADS, PP ? 4 : 2 ) void k_analyze_mr( AnalyzeParams p, MrPlan pl )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int C = pl.C, W = p.window_size, hop = p.hop, dft = 2 * C;
	const bool kc_lds = pl.kc_lds != 0;
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	const MrLds l = mr_carve<PP>( smem, p.tw, C, kc_lds, tid );
	cf * buf = l.buf;
	// the fused round trip's pre-pass inside the kernel (ping-pong sizes; beyond, where the frame leaves no LDS for it, k_phase_sums2 reads the PV
	// once more on the kernel's behalf): per bin the chain's phase increments summed in double, operation for operation what k_phase_sums2 does
	const bool sums = PP && p.sums != nullptr;
	double * s_sum = reinterpret_cast<double*>( l.state );                            // [C + 1] (PP; l.state is 8-byte aligned)
	float * s_prev = reinterpret_cast<float*>( s_sum + ( PP ? C + 1 : 0 ) );          // [C + 1]
	bool bad = false;
	for( int k = tid; k <= C; k += MR_THREADS )
		{
		s_prev[k] = 0.0f;                                                              // AudioPV.cpp:44
		if( PP ) s_sum[k] = 0.0;
		if( kc_lds )
			{
			const v4f_t v = mr_kc_value( k, p.tw2[k], p.sample_rate, p.analysis_rate, dft );
			cf * t = reinterpret_cast<cf*>( l.kc ) + 2 * k;
			t[0] = mk( v.x, v.y ); t[1] = mk( v.z, v.w );
			}
		}
	__syncthreads();
	if( s_cancel ) return;

	const int64_t chain = blockIdx.x;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                        // phase_vocoder.cpp:37
	constexpr int U = 4;                                                               // memory reads of U loop steps are requested together
	const cf * Z = l.buf2;                                                             // where the last spectrum stands (none yet: the first frame goes into buf)

	for( int64_t t = ( t0 > 0 ? t0 - 1 : t0 ); t < t1; ++t )                           // (the frame before the chain only lends its phases: phase_vocoder.cpp:45)
		{
		const bool emit = t >= t0;
		// the windowed frame as C complex points z[i] = ( x[2i], x[2i+1] ), zero beyond the window and outside the signal (AudioPV.cpp:52-65)
		const int64_t start = int64_t( hop ) * t - W / 2;
		// PP: the frame goes into the buffer the previous frame's spectrum is NOT in -- the threads still reading that spectrum (bins) and the
		// ones already loading the next frame do not meet, and the barrier between the two phases goes
		cf * const in = PP ? ( Z == buf ? l.buf2 : buf ) : buf;
		cf * const other = PP ? ( in == buf ? l.buf2 : buf ) : buf;
		for( int i0 = tid; i0 < C; i0 += U * MR_THREADS )
			{
			float a0[U], a1[U], w0[U], w1[U];
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int i = i0 + MR_THREADS * u, s0 = 2 * i, s1 = 2 * i + 1;
				const int64_t p0 = start + s0, p1 = start + s1;
				const bool ok0 = i < C && s0 < W && p0 >= 0 && p0 < p.n, ok1 = i < C && s1 < W && p1 >= 0 && p1 < p.n;
				a0[u] = ok0 ? x[p0] : 0.0f; w0[u] = ok0 ? p.window[s0] : 0.0f;
				a1[u] = ok1 ? x[p1] : 0.0f; w1[u] = ok1 ? p.window[s1] : 0.0f;
				}
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int i = i0 + MR_THREADS * u;
				if( i < C ) in[PAD( i )] = mk( a0[u] * w0[u], a1[u] * w1[u] );            // AudioPV.cpp:60 (0 x 0 outside)
				}
			}
		__syncthreads();
		Z = mr_fft<PP, BIG>( in, other, l.tw, pl, tid );

		// the real transform's bins from the half-size one, each phase-vocoded (AudioPV.cpp:69-73)
		MF * row = p.out + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		const cf z0 = Z[PAD( 0 )];
		for( int k0 = tid; k0 <= C; k0 += U * MR_THREADS )
			{
			v4f_t kc[U];
			#pragma unroll
			for( int u = 0; u < U; ++u ) kc[u] = mr_kc( l, kc_lds, p.tw2, min( k0 + MR_THREADS * u, C ), p.sample_rate, p.analysis_rate, dft );
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
					const cf zk = Z[PAD( k )], zm = Z[PAD( C - k )];
					const float ax = 0.5f * ( zk.x + zm.x ), ay = 0.5f * ( zk.y - zm.y );
					const float dx = zk.x - zm.x, dy = zk.y + zm.y;
					re = ax + 0.5f * __builtin_fmaf( kc[u].x, dy, kc[u].y * dx );
					im = ay - 0.5f * __builtin_fmaf( kc[u].x, dx, -( kc[u].y * dy ) );
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
		if constexpr( !PP ) __syncthreads();                                            // (one buffer: the next frame overwrites what the bins read)
		}
	if( PP && sums )
		{
		double * dst = p.sums + chain * ( C + 1 );
		for( int k = tid; k <= C; k += MR_THREADS ) { const double ph = s_sum[k]; dst[k] = ( __builtin_fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph ); }   // (own bins: no barrier)
		const bool any_bad = __any( bad );
		if( p.nan_out && ( tid & 63 ) == 0 )
			{
			if( chain == 0 && tid == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }   // (no clearing pass: set = equal to this launch's epoch)
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
		}
	}

// ---- PV::convert_to_audio (Conversions/AudioPV.cpp:86-139): one block per chain, from the carries of the common pre-pass ------------------
template<bool PP, bool BIG>
__global__ __launch_bounds__( MR_THREADS, PP ? 4 : 2 ) void k_synthesize_mr( SynthParams p, MrPlan pl )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int C = pl.C, W = p.window_size, hop = p.hop;
	const int wpad = ( W + 3 ) & ~3;
	const bool kc_lds = pl.kc_lds != 0, win_lds = pl.win_lds != 0;
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	const MrLds l = mr_carve<PP>( smem, p.tw, C, kc_lds, tid );
	cf * buf = l.buf;
	double * s_ph = reinterpret_cast<double*>( l.state );                             // [C + 2] (l.state is 8-byte aligned)
	// the overlap-add ring: [wpad] floats behind the running phases, or (MrPlan::ring_ws) this block's own stretch of the workspace -- it stays in the CU's L1 / the
	// XCD's L2, and the block barriers order its accesses exactly as they order the LDS ring's (k_synthesize_big: BigSynthExtra::ring_ws)
	float * const lds_ring = reinterpret_cast<float*>( s_ph + ( C + 2 ) );
	float * ring = pl.ring_ws ? p.ring_ws + int64_t( blockIdx.x ) * wpad : lds_ring;
	float * s_win = pl.ring_ws ? lds_ring : lds_ring + wpad;                          // [wpad], if win_lds: the scaled window (AudioPV.cpp:102)
	// (either an LDS read or a global one, in arms of their own: `c ? lds[i] : mem[i]` becomes one FLAT load through a selected pointer)
	auto win = [&]( int i ) { float v; if( win_lds ) { v = s_win[i]; asm volatile( "" : "+v"( v ) ); } else v = p.window[i] * p.window_scale; return v; };
	for( int i = tid; i < W; i += MR_THREADS ) { ring[i] = 0.0f; if( win_lds ) s_win[i] = p.window[i] * p.window_scale; }
	if( kc_lds ) for( int k = tid; k <= C; k += MR_THREADS ) reinterpret_cast<cf*>( l.kc )[2 * k] = p.tw2[k];   // (only the split twiddle is wanted here)

	const int64_t chain = blockIdx.x;
	const int channel = int( chain / p.chains_per_channel );
	const int chain_in_channel = int( chain % p.chains_per_channel );
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	float * out = p.out + int64_t( channel ) * p.out_len;
	float * head = p.head + chain * p.head_len;
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;                          // first sample this chain touches
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;   // before this: shared with the previous chain
	for( int k = tid; k <= C; k += MR_THREADS ) s_ph[k] = p.carry[chain * ( C + 1 ) + k];   // the running phase (AudioPV.cpp:105) on entry to the chain
	__syncthreads();
	if( s_cancel ) return;
	auto w2_of = [&]( int k ) { cf v; if( kc_lds ) { v = reinterpret_cast<const cf*>( l.kc )[2 * k]; asm volatile( "" : "+v"( v ) ); } else v = p.tw2[k]; return v; };
	constexpr int U = 4;

	int ring_base = 0;                                                                 // ring[ring_base] <-> absolute sample `pos`
	int64_t pos = chain_start;
	for( int64_t t = t0; t < t1; ++t )
		{
		// inverse phase vocoder per bin (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61) -> X[0 .. C] in LDS
		const MF * row = p.pv + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		for( int k0 = tid; k0 <= C; k0 += U * MR_THREADS )
			{
			MF mfs[U];
			#pragma unroll
			for( int u = 0; u < U; ++u ) mfs[u] = row[min( k0 + MR_THREADS * u, C )];
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
		__syncthreads();
		// merge X[0 .. C] into the half-size spectrum Z[k] = A[k] + i B[k], stored conjugated so that the forward transform evaluates the
		// inverse one ( ifft( Z ) = conj( fft( conj Z ) ) ); c2r ignores Im X[0] and Im X[C].  In place: a thread owns bin k AND its mirror
		auto merge = [&]( int k, cf xk, cf xm, cf w2q ) -> cf
			{
			if( k == 0 ) { xk.y = 0.0f; xm.y = 0.0f; }
			const float ax = xk.x + xm.x, ay = xk.y - xm.y;                             // A = X[k] + conj X[C-k]
			const float dx = xk.x - xm.x, dy = xk.y + xm.y;                             // B = ( X[k] - conj X[C-k] ) exp( +2 pi i k / N )
			const float c = w2q.x, s = -w2q.y;
			const float bx = __builtin_fmaf( c, dx, -( s * dy ) ), by = __builtin_fmaf( c, dy, s * dx );
			return mk( ax - by, -( ay + bx ) );
			};
		for( int k0 = tid; 2 * k0 <= C; k0 += U * MR_THREADS )
			{
			cf wa[U], wb[U];
			#pragma unroll
			for( int u = 0; u < U; ++u ) { const int k = min( k0 + MR_THREADS * u, C / 2 ); wa[u] = w2_of( k ); wb[u] = w2_of( C - k ); }
			#pragma unroll
			for( int u = 0; u < U; ++u )
				{
				const int k = k0 + MR_THREADS * u;
				if( 2 * k > C ) continue;
				const cf xk = buf[PAD( k )], xm = buf[PAD( C - k )];
				const cf zk = merge( k, xk, xm, wa[u] );
				if( k != 0 && 2 * k != C ) buf[PAD( C - k )] = merge( C - k, xm, xk, wb[u] );
				buf[PAD( k )] = zk;
				}
			}
		__syncthreads();
		const cf * G = mr_fft<PP, BIG>( buf, l.buf2, l.tw, pl, tid );
		// G = fft( conj Z ):  x[2n] = G[n].x, x[2n+1] = -G[n].y (AudioPV.cpp:122; samples from W on are discarded); window, accumulate (:133-134)
		for( int n = tid; 2 * n < W; n += MR_THREADS )
			{
			const cf g = G[PAD( n )];
			int i0 = ring_base + 2 * n; if( i0 >= W ) i0 -= W;
			ring[i0] += g.x * win( 2 * n );
			if( 2 * n + 1 < W )
				{
				int i1 = i0 + 1; if( i1 >= W ) i1 -= W;
				ring[i1] += ( -g.y ) * win( 2 * n + 1 );
				}
			}
		__syncthreads();
		// the oldest `hop` samples are complete as far as this chain is concerned: emit and clear them
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
		// (no barrier: the ring is next touched behind the barriers of the next frame's passes)
		pos += hop;
		ring_base = ( hop < W ) ? ring_base + hop : 0;
		if( ring_base >= W ) ring_base -= W;
		}
	// what is left in the ring (partial sums the next chain's head completes); a channel's last chain zero-fills to the end of the output
	// (Audio( format ) is zero-initialised, AudioPV.cpp:95)
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
