// pv_kernels_big.h -- Audio::convert_to_PV / PV::convert_to_audio for power-of-two-rich dft sizes ABOVE 16384 (32768, 65536 ... 2^20, 24576 ...),
// which FFTW plans like any other size (reference: FFTHelper.cpp:16-26; Audio.h:158-163 constrains nothing) and which ran as direct sums here until
// round 5 (pv_kernels_any.h: O( window x bins ) per frame, ~0.4 s for 8 ch x 60 s at dft 32768).
//
// Half the size C = C1 x C2 with C2 = 4096 ( 2048, 1024 ) complex points -- what a block transforms in LDS -- and C1 = 2 ... 256; round 6 (MIXED): any C2
// between 256 and 4096 that is a product of 2 ... 13, for the sizes whose half holds less than 2^10 as a power of two (20000, 44100, 48000, 100000 ...:
// bs_plan.h), with pv_kernels_mr.h's odd-radix passes and the threads' bins guarded (C2 < 512 Q).  Decimation in frequency:
//     Z[ k1 + C1 k2 ] = fft_C2( y_k1 )[ k2 ],     y_k1[ n2 ] = w_C^( n2 k1 ) sum_n1 z[ n2 + C2 n1 ] w_C1^( n1 k1 )
// and a frame is shorter than C2 complex points unless the window is above 2 C2 samples, so the inner sum is one term: the zero-padded transform IS C1
// separate C2-point transforms of the frame times a twiddle (longer windows: the sum over the segments that are not zero).
// The real transform's split pairs bin k with C - k, i.e. residue k1 with C1 - k1:  C - ( k1 + C1 k2 ) = ( C1 - k1 ) + C1 ( C2 - 1 - k2 ).  So the unit
// of work is a PAIR of residues { k1, C1 - k1 } of one chain of frames ( { 0 } and { C1 / 2 } pair with themselves ): a block of 512 threads holds the
// pair's two transforms in two LDS buffers (in place, radix 8 / 4 / 2 passes of pv_kernels_bs.h / _mr.h) and the pair's bins' state -- previous phases,
// running phases, per-bin constants -- in registers: bin k2 = tid + 512 q of either residue belongs to thread tid for the whole chain.
// Synthesis: G[ n ] = sum_k1 w_C^( n k1 ) fft_C2( Zc[ k1 + C1 . ] )[ n mod C2 ] -- every unit adds ITS residues' share into an overlap-add ring of
// its own and writes a partial output stream; k_big_reduce adds the units' streams (and the chains' overlapping heads) in a fixed order.
// Rows of a PV are touched with a stride of C1 bins by a unit; the grid is ordered so that the units of one chain run on one XCD at about the same
// time (blockIdx % 8 = chain % 8) and their partial lines meet in that XCD's L2.
#pragma once
#include "pv_kernels_bs.h"

namespace flanhip {

// (BigPlan, big_make_plan: bs_plan.h -- host arithmetic, also compiled by the CPU tests)

// LDS of one block (bytes): [twiddles C2 cf][two frames padded_len( C2 + 1 ) cf][synthesis: ring]
inline size_t big_analyze_lds( int C2 ) { return size_t( C2 ) * 8 + 2 * size_t( padded_len( C2 + 1 ) ) * 8; }
inline size_t big_synth_lds( int C2, int W ) { return big_analyze_lds( C2 ) + size_t( ( W + 3 ) & ~3 ) * 4; }

// The compiler hoists every loop-invariant address of the unrolled per-bin loops out of the frame loop (Q x { row index, two LDS slots, table index }
// per residue: ~60 registers that then spill); an index that passes through this cannot be proven invariant and is recomputed where it is used.
__device__ __forceinline__ int big_opaque( int v ) { asm volatile( "" : "+v"( v ) ); return v; }

// blockIdx -> ( chain, unit ): [ chain / 8 ][ unit ][ chain % 8 ]
__device__ __forceinline__ bool big_block( int64_t chains, int P, int64_t & chain, int & unit )
	{
	const int64_t b = blockIdx.x;
	chain = ( b / ( 8 * int64_t( P ) ) ) * 8 + ( b & 7 );
	unit = int( ( b >> 3 ) % P );
	return chain < chains;
	}
inline int64_t big_blocks( int64_t chains, int P ) { return ( ( chains + 7 ) / 8 ) * 8 * int64_t( P ); }

// forward transform of the C2 points in `a`, in place (natural order in and out); points from `limit` on are zero on entry and need not have been written.
// Powers of two: the chirp-z kernels' first pass (radix 8, reads no point from `limit` on), then 8 / 4 / 2; MIXED: the tail is zeroed, then every radix of the plan
// (MIXED = 1: 2 ... 7; 2: 11 and 13 as well -- their butterflies of 2 x 13 points spill ~450 bytes per thread, which only the plans that hold one pay)
template<int MIXED> __device__ __forceinline__ void big_fft( cf * a, const cf * tw, const BigFft & f, int limit, int tid )
	{
	int NS = 1, i = 0;
	if constexpr( MIXED )
		{
		for( int j = limit + tid; j < f.M; j += MR_THREADS ) a[PAD( j )] = mk( 0.0f, 0.0f );
		__syncthreads();
		}
	else
		{
		bs_pass0<false, false>( a, a, f.M, limit, nullptr, nullptr, tid );
		NS = 8; i = 1;
		}
	for( ; i < f.npass; ++i )
		{
		const int r = f.radix[i];
		bool done = true;
		switch( r )
			{
			case 8:  mr_pass<8, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid ); break;
			case 4:  mr_pass<4, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid ); break;
			case 2:  mr_pass<2, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid ); break;
			default: done = false; break;
			}
		if constexpr( MIXED > 0 )
			{
			if( !done )
				{
				done = true;
				switch( r )
					{
					case 3:  mr_pass<3, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid ); break;
					case 5:  mr_pass<5, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid ); break;
					case 7:  mr_pass<7, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid ); break;
					default: done = false; break;
					}
				}
			}
		if constexpr( MIXED > 1 )
			{
			if( !done )
				{
				if( r == 11 ) mr_pass<11, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid );
				else mr_pass<13, false>( a, a, tw, f.M, NS, f.magic[i], f.stride[i], tid );
				}
			}
		NS *= r;
		}
	}

// ---- Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78) -------------------------------------------------------------------------------
template<int Q, int MIXED = 0>                                                       // Q = ceil( C2 / 512 ) (2, 4, 8): bins of a residue per thread
__global__ __launch_bounds__( MR_THREADS, 2 ) void k_analyze_big( AnalyzeParams p, BigPlan pl )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int C = pl.C, C1 = pl.C1, C2 = pl.C2, W = p.window_size, hop = p.hop, dft = 2 * C;
	int64_t chain; int unit;
	if( !big_block( int64_t( p.chains_per_channel ) * p.num_channels, pl.P, chain, unit ) ) return;
	__shared__ int s_cancel;                                                            // (one thread's reading for the whole block: the word may rise between two wavefronts' reads)
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	cf * s_tw = reinterpret_cast<cf*>( smem );
	cf * bufA = s_tw + C2, * bufB = bufA + padded_len( C2 + 1 );
	for( int j = tid; j < C2; j += MR_THREADS ) s_tw[j] = p.tw[int64_t( j ) * C1];     // exp( -2 pi i j / C2 )

	const int ka = unit, kb = ( C1 - unit ) % C1;
	const bool paired = ka != kb;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                        // phase_vocoder.cpp:37

	// the thread's bins and what it keeps for them: bin kA[q] = ka + C1 ( tid + 512 q ) of residue a, kB[q] likewise (paired units)
	float prevA[Q], prevB[Q], prevC = 0.0f;
	#pragma unroll
	for( int q = 0; q < Q; ++q ) { prevA[q] = 0.0f; prevB[q] = 0.0f; }                   // AudioPV.cpp:44
	const v4f_t kcC = mr_kc_value( C, p.tw2[C], p.sample_rate, p.analysis_rate, dft );
	__syncthreads();
	if( s_cancel ) return;

	for( int64_t t = ( t0 > 0 ? t0 - 1 : t0 ); t < t1; ++t )                           // (the frame before the chain only lends its phases: phase_vocoder.cpp:45)
		{
		const bool emit = t >= t0;
		const int64_t start = int64_t( hop ) * t - W / 2;
		// y_k1[ n2 ] for both residues, n2 < limit (AudioPV.cpp:52-65 for the samples)
		for( int n2 = tid; n2 < pl.limit; n2 += MR_THREADS )
			{
			cf sa = mk( 0.0f, 0.0f ), sb = mk( 0.0f, 0.0f );
			for( int n1 = 0; n1 < pl.N1; ++n1 )
				{
				const int i = n2 + C2 * n1, s0 = 2 * i, s1 = 2 * i + 1;
				const int64_t p0 = start + s0, p1 = start + s1;
				const bool ok0 = s0 < W && p0 >= 0 && p0 < p.n, ok1 = s1 < W && p1 >= 0 && p1 < p.n;
				const cf z = mk( ok0 ? x[p0] * p.window[s0] : 0.0f, ok1 ? x[p1] * p.window[s1] : 0.0f );   // AudioPV.cpp:60
				if( n1 == 0 ) { sa = z; sb = z; }
				else
					{
					sa = cadd( sa, cmul( z, p.tw[int64_t( ( n1 * ka ) % C1 ) * C2] ) );
					sb = cadd( sb, cmul( z, p.tw[int64_t( ( n1 * kb ) % C1 ) * C2] ) );
					}
				}
			bufA[PAD( n2 )] = ka ? cmul( sa, p.tw[n2 * ka] ) : sa;                          // ( n2 k1 < C2 C1 = C )
			if( paired ) bufB[PAD( n2 )] = cmul( sb, p.tw[n2 * kb] );
			}
		__syncthreads();
		big_fft<MIXED>( bufA, s_tw, pl.fft, pl.limit, tid );
		if( paired ) big_fft<MIXED>( bufB, s_tw, pl.fft, pl.limit, tid );

		// the real transform's bins, each phase-vocoded (AudioPV.cpp:69-73): Z[k] from this residue, Z[C - k] from the other (see the header)
		MF * row = p.out + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		const cf * const mirror_of_a = paired ? bufB : bufA;
		auto bin = [&]( cf zk, cf zm, float & prev, int k )
			{
			const v4f_t kc = mr_kc_value( k, p.tw2[k], p.sample_rate, p.analysis_rate, dft );
			const float ax = 0.5f * ( zk.x + zm.x ), ay = 0.5f * ( zk.y - zm.y );
			const float dx = zk.x - zm.x, dy = zk.y + zm.y;
			const float re = ax + 0.5f * __builtin_fmaf( kc.x, dy, kc.y * dx );
			const float im = ay - 0.5f * __builtin_fmaf( kc.x, dx, -( kc.y * dy ) );
			const MF mf = phase_vocode_bin( re, im, prev, kc.z, kc.w, p.analysis_rate, use_wrapping );
			if( emit ) row[k] = mf;
			};
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			__builtin_amdgcn_sched_barrier( 0 );                                            // (bin after bin: interleaved, the Q bodies' temporaries spill)
			const int k2 = big_opaque( tid ) + MR_THREADS * q;
			if( MIXED && k2 >= C2 ) continue;
			const cf za = bufA[PAD( k2 )];
			if( ka == 0 )
				{
				if( k2 == 0 )
					{
					// bins 0 and C: exact reals (r2c)
					const v4f_t kc0 = mr_kc_value( 0, p.tw2[0], p.sample_rate, p.analysis_rate, dft );
					const MF m0 = phase_vocode_bin( za.x + za.y, 0.0f, prevA[q], kc0.z, kc0.w, p.analysis_rate, use_wrapping );
					const MF mc = phase_vocode_bin( za.x - za.y, 0.0f, prevC, kcC.z, kcC.w, p.analysis_rate, use_wrapping );
					if( emit ) { row[0] = m0; row[C] = mc; }
					}
				else bin( za, bufA[PAD( C2 - k2 )], prevA[q], C1 * k2 );
				}
			else
				{
				bin( za, mirror_of_a[PAD( C2 - 1 - k2 )], prevA[q], ka + C1 * k2 );
				if( paired ) bin( bufB[PAD( k2 )], bufA[PAD( C2 - 1 - k2 )], prevB[q], kb + C1 * k2 );
				}
			}
		__syncthreads();                                                                  // (the next frame overwrites what the bins read)
		}
	}

// ---- PV::convert_to_audio (Conversions/AudioPV.cpp:86-139) ---------------------------------------------------------------------------------
struct BigSynthExtra
	{
	float * part_out;         // [P][ch][out_len]   a unit's share of the output
	float * part_head;        // [P][chains][head_len]
	float * ring_ws;          // [P][chains][wpad] or null: the units' overlap-add rings in the workspace, for windows whose ring does not fit the LDS beside the
	                          // transforms (above ~14 k samples: convert_to_PV( 32768, 8192, 32768 ), a PaulStretch-style call, ran the direct sums until round 6).
	                          // A ring is its block's alone and stays in that CU's L1 / the XCD's L2; the block barriers order its accesses like the LDS ring's
	};

template<int Q, int MIXED = 0>
__global__ __launch_bounds__( MR_THREADS, 2 ) void k_synthesize_big( SynthParams p, BigPlan pl, BigSynthExtra e )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int C = pl.C, C1 = pl.C1, C2 = pl.C2, W = p.window_size, hop = p.hop;
	const int wpad = ( W + 3 ) & ~3;
	const int64_t chains = int64_t( p.chains_per_channel ) * p.num_channels;
	int64_t chain; int unit;
	if( !big_block( chains, pl.P, chain, unit ) ) return;
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	cf * s_tw = reinterpret_cast<cf*>( smem );
	cf * bufA = s_tw + C2, * bufB = bufA + padded_len( C2 + 1 );
	float * ring = e.ring_ws ? e.ring_ws + ( int64_t( unit ) * chains + chain ) * wpad : reinterpret_cast<float*>( bufB + padded_len( C2 + 1 ) );   // [wpad]
	for( int j = tid; j < C2; j += MR_THREADS ) s_tw[j] = p.tw[int64_t( j ) * C1];
	for( int i = tid; i < wpad; i += MR_THREADS ) ring[i] = 0.0f;

	const int ka = unit, kb = ( C1 - unit ) % C1;
	const bool paired = ka != kb;
	const int channel = int( chain / p.chains_per_channel );
	const int chain_in_channel = int( chain % p.chains_per_channel );
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	float * out = e.part_out + ( int64_t( unit ) * p.num_channels + channel ) * p.out_len;
	float * head = e.part_head + ( int64_t( unit ) * chains + chain ) * p.head_len;
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;

	// running phases (AudioPV.cpp:105) of the thread's bins, from the chain's carry
	double phA[Q], phB[Q], phC = 0.0;
	const double * carry = p.carry + chain * ( C + 1 );
	#pragma unroll
	for( int q = 0; q < Q; ++q )
		{
		const int k2 = ( MIXED && tid + MR_THREADS * q >= C2 ) ? 0 : tid + MR_THREADS * q;     // (bins past the residue's end: never used)
		phA[q] = carry[ka + C1 * k2];
		phB[q] = carry[kb + C1 * k2];
		}
	if( ka == 0 && tid == 0 ) phC = carry[C];
	__syncthreads();
	if( s_cancel ) return;

	auto polar_of = [&]( MF mf, double & ph ) -> cf                                    // AudioPV.cpp:117-120, phase_vocoder.cpp:55-61
		{
		const double term = double( div_c( mf.f, p.ar_div ) * FLANHIP_PI2_F );
		double phase = ph + term;
		phase = ( __builtin_fabs( phase ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( phase ) : fold_phase_any( phase );
		ph = phase;
		const float th = float( phase );
		float sn, cs;
		if( __builtin_fabsf( th ) < FLANHIP_SINCOS_FAST_LIMIT ) sincos_fast( th, sn, cs );
		else { const float2 sc = sincos_wide( th ); sn = sc.x; cs = sc.y; }
		return mk( mf.m * cs, mf.m * sn );
		};
	// Zc[k] = A[k] + i B[k], conjugated (k_synthesize_mr: merge), from X[k] and X[C - k]
	auto merge = [&]( int k, cf xk, cf xm ) -> cf
		{
		const cf w2q = p.tw2[k];
		if( k == 0 ) { xk.y = 0.0f; xm.y = 0.0f; }
		const float ax = xk.x + xm.x, ay = xk.y - xm.y;
		const float dx = xk.x - xm.x, dy = xk.y + xm.y;
		const float c = w2q.x, s = -w2q.y;
		const float bx = __builtin_fmaf( c, dx, -( s * dy ) ), by = __builtin_fmaf( c, dy, s * dx );
		return mk( ax - by, -( ay + bx ) );
		};

	int ring_base = 0;
	int64_t pos = chain_start;
	for( int64_t t = t0; t < t1; ++t )
		{
		const MF * row = p.pv + ( int64_t( channel ) * p.F + t ) * ( C + 1 );
		MF ma[Q], mb[Q];
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			int k2 = big_opaque( tid ) + MR_THREADS * q;
			if( MIXED && k2 >= C2 ) k2 = 0;
			ma[q] = row[ka + C1 * k2]; if( paired ) mb[q] = row[kb + C1 * k2];
			}
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			__builtin_amdgcn_sched_barrier( 0 );                                            // (bin after bin: interleaved, the Q bodies' temporaries spill)
			const int k2 = big_opaque( tid ) + MR_THREADS * q;
			if( MIXED && k2 >= C2 ) continue;
			bufA[PAD( k2 )] = polar_of( ma[q], phA[q] );
			__builtin_amdgcn_sched_barrier( 0 );
			if( paired ) bufB[PAD( k2 )] = polar_of( mb[q], phB[q] );
			}
		if( ka == 0 && tid == 0 ) bufA[PAD( C2 )] = polar_of( row[C], phC );               // X[C] in the slot behind the residue's points
		__syncthreads();
		// merge in place: a thread owns bin ( a, k2 ) AND its mirror
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			__builtin_amdgcn_sched_barrier( 0 );
			const int k2 = big_opaque( tid ) + MR_THREADS * q;
			if( MIXED && k2 >= C2 ) continue;
			if( ka == 0 )
				{
				// residue 0 mirrors itself: k = C1 k2 <-> C - k = C1 ( C2 - k2 );  k2 = 0: X[0] with X[C]
				const int m2 = C2 - k2;
				if( k2 == 0 ) bufA[PAD( 0 )] = merge( 0, bufA[PAD( 0 )], bufA[PAD( C2 )] );
				else if( k2 <= m2 )
					{
					const cf xk = bufA[PAD( k2 )], xm = bufA[PAD( m2 )];
					bufA[PAD( k2 )] = merge( C1 * k2, xk, xm );
					if( k2 != m2 ) bufA[PAD( m2 )] = merge( C1 * m2, xm, xk );
					}
				}
			else if( paired )
				{
				const int m2 = C2 - 1 - k2;
				const cf xk = bufA[PAD( k2 )], xm = bufB[PAD( m2 )];
				bufA[PAD( k2 )] = merge( ka + C1 * k2, xk, xm );
				bufB[PAD( m2 )] = merge( kb + C1 * m2, xm, xk );
				}
			else
				{
				// the middle residue mirrors itself: ( k2, C2 - 1 - k2 ); an odd C2 (MIXED) has a bin that is its own mirror there: C / 2
				const int m2 = C2 - 1 - k2;
				if( k2 <= m2 )
					{
					const cf xk = bufA[PAD( k2 )], xm = bufA[PAD( m2 )];
					bufA[PAD( k2 )] = merge( ka + C1 * k2, xk, xm );
					if( k2 != m2 ) bufA[PAD( m2 )] = merge( ka + C1 * m2, xm, xk );
					}
				}
			}
		__syncthreads();
		big_fft<MIXED>( bufA, s_tw, pl.fft, C2, tid );
		if( paired ) big_fft<MIXED>( bufB, s_tw, pl.fft, C2, tid );
		// this unit's share of G[n] = fft_C( conj Z )[n]:  x[2n] = G[n].x, x[2n+1] = -G[n].y (AudioPV.cpp:122); window, accumulate (:133-134)
		for( int n = tid; 2 * n < W; n += MR_THREADS )
			{
			const int r = MIXED ? n % C2 : n & ( C2 - 1 );
			cf gsum = bufA[PAD( r )];
			if( ka ) gsum = cmul( gsum, p.tw[( int64_t( n ) * ka ) % C] );
			if( paired ) gsum = cadd( gsum, cmul( bufB[PAD( r )], p.tw[( int64_t( n ) * kb ) % C] ) );
			int i0 = ring_base + 2 * n; if( i0 >= W ) i0 -= W;
			ring[i0] += gsum.x * ( p.window[2 * n] * p.window_scale );
			if( 2 * n + 1 < W )
				{
				int i1 = i0 + 1; if( i1 >= W ) i1 -= W;
				ring[i1] += ( -gsum.y ) * ( p.window[2 * n + 1] * p.window_scale );
				}
			}
		__syncthreads();
		for( int i = tid; i < hop; i += MR_THREADS )
			{
			float v = 0.0f;
			if( i < W )
				{
				int j = ring_base + i; if( j >= W ) j -= W;
				v = ring[j]; ring[j] = 0.0f;
				}
			const int64_t a = pos + i;
			if( a < own_start ) head[a - chain_start] = v;
			else if( a >= 0 && a < p.out_len ) out[a] = v;
			}
		pos += hop;
		ring_base = ( hop < W ) ? ring_base + hop : 0;
		if( ring_base >= W ) ring_base -= W;
		__syncthreads();
		}
	const int64_t ring_end = pos + ( hop < W ? W - hop : 0 );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	for( int64_t a = pos + tid; a < flush_end; a += MR_THREADS )
		{
		float v = 0.0f;
		if( a < ring_end )
			{
			int j = ring_base + int( a - pos ); if( j >= W ) j -= W;
			v = ring[j];
			}
		if( a < own_start ) head[a - chain_start] = v;
		else if( a >= 0 && a < p.out_len ) out[a] = v;
		}
	}

// out[c][a] = sum over the units, in ascending order, of ( the unit's stream + the head of the chain that starts under a ): one thread per sample
__global__ __launch_bounds__( 256 ) void k_big_reduce( SynthParams p, BigPlan pl, BigSynthExtra e )
	{
	const int64_t i = int64_t( blockIdx.x ) * 256 + threadIdx.x;
	const int64_t total = int64_t( p.num_channels ) * p.out_len;
	if( i >= total ) return;
	const int channel = int( i / p.out_len );
	const int64_t a = i - int64_t( channel ) * p.out_len;
	const int64_t chains = int64_t( p.chains_per_channel ) * p.num_channels;
	// the chain whose head region [ chain_start, chain_start + head_len ) may hold a (chains are L hop samples apart, head_len <= L hop)
	const int64_t span = int64_t( p.L ) * p.hop;
	const int64_t j = ( a + p.window_size / 2 ) / span;
	const int64_t cs = span * j - p.window_size / 2;
	const bool in_head = j >= 1 && j < p.chains_per_channel && a - cs < p.head_len;
	float sum = 0.0f;
	for( int u = 0; u < pl.P; ++u )
		{
		float v = e.part_out[( int64_t( u ) * p.num_channels + channel ) * p.out_len + a];
		if( in_head ) v += e.part_head[( int64_t( u ) * chains + int64_t( channel ) * p.chains_per_channel + j ) * p.head_len + ( a - cs )];   // tail + head, as k_ola_fixup
		sum = u ? sum + v : v;
		}
	p.out[i] = sum;
	}

} // namespace flanhip
