// pv_kernels_sub.h -- dft 512, 256 and 128 (C = 256 / 128 / 64 complex points per frame): SEVERAL CHAINS PER WAVEFRONT (round 6).
//
// pv_kernels_v3.h walks one chain per wavefront with E = C / 64 points per lane: 4 at dft 512 (four dependent LDS exchanges of four points, two bin pairs
// per lane and frame -- 1.8 x the dft 1024 kernel's cost per bin), 2 at dft 256 (which therefore ran the generic kernels).  Here a lane always holds E = 8
// points -- the dft 1024 kernel's instruction mix: 8 x 8 x ( C / 64 ), two exchanges, four bin pairs per lane and frame -- and a chain takes LP = C / 8
// lanes: 32 at dft 512, 16 at dft 256, 8 at dft 128 (added at the round's end: the generic kernels ran it at 50 G bins/s, these at 137).  The G = 64 / LP lane groups of a wavefront walk G CONSECUTIVE CHAINS of one channel side by side, frame for
// frame; nothing crosses between the groups (a chain is a chain: previous phases, running phases, overlap-add accumulator are a lane's own), so there are
// no cross-lane operations in the frame loop beyond the transform's own exchanges, each group in its own LDS buffer.  What differs between the groups of a
// wavefront is data, not control: every group walks L + 1 (L) iterations; a channel's first chain has no halo frame (its halo iteration's phases are
// replaced by zeros, AudioPV.cpp:44), its last chain may be short (its surplus iterations load clamped rows and store nothing), spare groups of a channel's
// last wavefront repeat the last chain without storing.
// Analysis: Conversions/AudioPV.cpp:12-78, phase_vocoder.cpp:37-52; synthesis: AudioPV.cpp:86-139, phase_vocoder.cpp:55-61.  Carries: the analysis leaves chain
// sums and one total per block (a group = the block's 8 / 16 chains), a scan over the GROUP totals gives every group's carry and the synthesis block's prologue
// works out its chains' carries from that and the chain sums (k_synthesize_v2's form; a PV from elsewhere: k_sums_and_groups); few chains per channel: k_phase_scan2
// over the chain sums.  The chains' overlaps: k_ola_fixup4.
#pragma once
#include "pv_kernels_v2.h"

namespace flanhip {

template<int LOG2C, int LP> struct SubLds
	{
	static constexpr int C = 1 << LOG2C;
	static_assert( C == 8 * LP && ( LP == 32 || LP == 16 || LP == 8 ), "eight points per lane; dft 512 / 256 / 128" );
	static constexpr int G = 64 / LP;                        // chains per wavefront
	static constexpr int R2 = C / 64;                        // radix of the last pass (4 / 2; 1 at dft 128: the pass only carries the elements back to their lanes), G butterflies per lane
	static constexpr int QS = LP + LP / 8;                   // slots between elements i and i + LP (one lane's consecutive points)
	static constexpr int TW1 = 0;                            // [7][8]        exp( -2 pi i r k / 64 )
	static constexpr int TW2 = TW1 + 56;                     // [R2 - 1][64]  exp( -2 pi i r j / C )
	static constexpr int W2 = TW2 + ( R2 - 1 ) * 64;         // [C/2]  exp( -+ 2 pi i k / 2C ) (analysis: halved)
	static constexpr int WIN = W2 + C / 2;                   // [2C floats]
	static constexpr int KC = WIN + C;                       // analysis: [C/2] float4 { binf(k), expected(k), binf(C-k), expected(C-k) }
	// highest slot: PAD( C ) = C + C / 8.  16 lanes per chain: two chains share a half-wavefront's LDS cycle, and with their buffers 16 slots (32 banks) apart
	// modulo 32 the scatter of the first pass ( 9 l + r ) lands on complementary banks
	static constexpr int BUF_LEN = LP == 16 ? 176 : C + C / 8 + 2;
	static constexpr int buf0( bool analysis ) { return analysis ? KC + C : KC; }
	static constexpr size_t bytes( int waves, bool analysis ) { return size_t( buf0( analysis ) + waves * G * BUF_LEN ) * 8; }
	};

template<int LOG2C, int LP, int NT> __device__ __forceinline__ void sub_load_twiddles( cf * s, const cf * tw, int tid )
	{
	using L = SubLds<LOG2C, LP>;
	for( int i = tid; i < 56; i += NT ) { const int r = i / 8 + 1, k = i % 8; s[L::TW1 + i] = tw[r * k * ( L::C / 64 )]; }
	for( int i = tid; i < ( L::R2 - 1 ) * 64; i += NT ) { const int r = i / 64 + 1, j = i % 64; s[L::TW2 + i] = tw[r * j]; }
	}

// forward transform of the group's C points: natural register layout in and out ( z[q] = element l + LP q ), through the group's own buffer
template<int LOG2C, int LP>
__device__ __forceinline__ void fft_sub( cf ( &z )[8], cf * buf, const cf * s, int l )
	{
	using L = SubLds<LOG2C, LP>;
	constexpr int QS = L::QS, R2 = L::R2, G = L::G;
	const int padl = l + ( l >> 3 );
	// pass 0: radix 8, inputs l + LP r, outputs out[ 8 l + r ]
	dft_reg<8>( z );
		{
		cf * wp = buf + 9 * l;
		#pragma unroll
		for( int r = 0; r < 8; ++r ) wp[r] = z[r];
		}
	wave_sync();
	// pass 1: radix 8, sub-transform length 8: butterfly l, inputs in[ l + LP r ], k = l mod 8, outputs out[ ( l - k ) 8 + k + 8 r ]
		{
		cf v[8], tw[7];
		const cf * rp = buf + padl;
		#pragma unroll
		for( int r = 0; r < 8; ++r ) v[r] = rp[QS * r];
		const int k = l & 7;
		const cf * tp = s + L::TW1 + k;
		#pragma unroll
		for( int r = 1; r < 8; ++r ) tw[r - 1] = tp[( r - 1 ) * 8];
		wave_sync();
		#pragma unroll
		for( int r = 1; r < 8; ++r ) v[r] = cmul( v[r], tw[r - 1] );
		dft_reg<8>( v );
		cf * wp = buf + 9 * ( l - k ) + k;
		#pragma unroll
		for( int r = 0; r < 8; ++r ) wp[9 * r] = v[r];
		}
	wave_sync();
	// pass 2: radix R2, sub-transform length 64: butterflies j = l + LP b, b < G; inputs in[ j + 64 r ], outputs out[ j + 64 r ] = element l + LP ( b + G r )
		{
		const cf * rp = buf + padl;
		const cf * tp = s + L::TW2 + l;
		#pragma unroll
		for( int b = 0; b < G; ++b )
			{
			cf v[R2];
			#pragma unroll
			for( int r = 0; r < R2; ++r ) v[r] = rp[QS * b + 72 * r];
			#pragma unroll
			for( int r = 1; r < R2; ++r ) v[r] = cmul( v[r], tp[LP * b + ( r - 1 ) * 64] );
			if constexpr( R2 > 1 ) dft_reg<R2>( v );
			#pragma unroll
			for( int r = 0; r < R2; ++r ) z[b + G * r] = v[r];
			}
		}
	wave_sync();
	}

// =================================================================================================================
// Audio::convert_to_PV, dft 512 / 256
// =================================================================================================================
#ifndef FLANHIP_SUB_ABL
#define FLANHIP_SUB_ABL 0      /* timing experiments only: 1 no store of bin C/2, 2 no MF stores, 4 plain instead of non-temporal MF stores */
#endif
template<int LOG2C, int LP, int WAVES, bool SUMS, int OCC, int NV = 4>
__global__ __launch_bounds__( 64 * WAVES, OCC ) void k_analyze_sub( AnalyzeParams p )
	{
	constexpr int ABL = FLANHIP_SUB_ABL;
	using L = SubLds<LOG2C, LP>;
	constexpr int C = L::C, E = 8, H = 4, G = L::G, NT = 64 * WAVES, QS = L::QS, NCH = WAVES * G;
	constexpr int NP = NV / 2;                                                  // bin pairs evaluated together as one vector stream
	static_assert( NV == 4 || NV == 8, "two or four pairs at a time" );
	using VB = FA<NV>;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 ), l = lane & ( LP - 1 ), g = lane / LP;
	const int W = p.window_size, hop = p.hop;
	const int cancel_word = cancel_peek( p.cancel );

	sub_load_twiddles<LOG2C, LP, NT>( s, p.tw, tid );
	for( int i = tid; i < C / 2; i += NT ) { const cf w = p.tw2[i]; s[L::W2 + i] = mk( 0.5f * w.x, 0.5f * w.y ); }
		{
		float * win = reinterpret_cast<float*>( s + L::WIN );
		for( int i = tid; i < 2 * C; i += NT ) win[i] = ( i < W ) ? p.window[i] : 0.0f;          // AudioPV.cpp:60,65
		v4f_t * kc = reinterpret_cast<v4f_t*>( s + L::KC );
		const float rdft = 1.0f / float( 2 * C );
		for( int k = tid; k < C / 2; k += NT )
			{
			const float bk = float( k ) * p.sample_rate * rdft, bm = float( C - k ) * p.sample_rate * rdft;      // PVBuffer.cpp:443-446
			kc[k] = v4f_t{ bk, div_c( bk, p.ar_div ) * FLANHIP_PI2_F, bm, div_c( bm, p.ar_div ) * FLANHIP_PI2_F };   // phase_vocoder.cpp:47
			}
		}
	__syncthreads();
	if( cancel_seen( cancel_word ) ) return;
	const cf * s_w2 = s + L::W2 + l;
	const cf * s_win = s + L::WIN + l;
	const v4f_t * s_kc = reinterpret_cast<const v4f_t*>( s + L::KC ) + l;
	cf * buf = s + L::buf0( true ) + ( wave * G + g ) * L::BUF_LEN;

	// a block: NCH consecutive chains of ONE channel
	const int groups = ( p.chains_per_channel + NCH - 1 ) / NCH;
	const int channel = int( blockIdx.x ) / groups, group = int( blockIdx.x ) % groups;
	const int cic_raw = group * NCH + wave * G + g;
	const bool active = cic_raw < p.chains_per_channel;
	const int cic = active ? cic_raw : p.chains_per_channel - 1;               // (spare groups repeat the channel's last chain and store nothing)
	const int64_t chain = int64_t( channel ) * p.chains_per_channel + cic;
	const int t0 = cic * p.L;
	const int t1 = int( min( int64_t( t0 ) + p.L, p.F ) );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const int n32 = int( p.n );
	// addresses: a scalar base per block + a 32-bit per-lane byte offset (everything a block touches lies within NCH L + 1 frames of tb0)
	const int tb0 = group * NCH * p.L - 1;
	const char * const xb = reinterpret_cast<const char*>( x ) + ( int64_t( hop ) * tb0 - W / 2 ) * 4;
	char * const rb = reinterpret_cast<char*>( p.out + ( int64_t( channel ) * p.F + tb0 ) * ( C + 1 ) );
	const unsigned l8 = 8u * unsigned( l );
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37
	const float k_ar = p.analysis_rate;
	const DivC k_ard = p.ar_div;
	const int padl = l + ( l >> 3 );
	const cf * mirror = buf + ( C + C / 8 ) - l - ( ( l + 7 ) >> 3 );         // mirror[-QS q] = slot PAD( C - l - LP q )
	struct __attribute__(( packed, aligned( 4 ) )) f2u { float x, y; };

	float prevk[H], prevm[H], prevx = 0.0f;
	#pragma unroll
	for( int q = 0; q < H; ++q ) { prevk[q] = 0.0f; prevm[q] = 0.0f; }        // AudioPV.cpp:44
	double sumk[SUMS ? H : 1], summ[SUMS ? H : 1], sumx = 0.0;
	if constexpr( SUMS )
		{
		#pragma unroll
		for( int q = 0; q < H; ++q ) { sumk[q] = 0.0; summ[q] = 0.0; }
		}
	unsigned mmax = 0u;
	cf ring = mk( 0.0f, 0.0f );                                               // Z[ C/2 ] of the chain's i-th iteration waits in lane i % LP of its group

	const bool w_whole = ( W & 127 ) == 0;
	auto frame_inside = [&]( int t ) { return w_whole && hop * t - W / 2 >= 0 && hop * t - W / 2 + 2 * C <= n32; };
	// iteration i: frame t0 - 1 + i (i = 0: the halo, of which only the phases are wanted -- for a channel's first chain not even those), clamped into the chain
	auto frame_of = [&]( int i ) { return min( max( t0 - 1 + i, 0 ), t1 - 1 ); };

	cf z[E];
	auto load_pair = [&]( int t, int q, auto fast_tag ) -> cf
		{
		constexpr bool FAST = decltype( fast_tag )::value;
		if constexpr( FAST )
			{
			const unsigned off = unsigned( hop * ( t - tb0 ) ) * 4u + l8;
			const f2u v = *reinterpret_cast<const f2u*>( xb + off + 8 * LP * q );
			return mk( v.x, v.y );
			}
		else
			{
			const int start = hop * t - W / 2;
			const int a0c = min( max( start + 2 * ( l + LP * q ), 0 ), n32 - 2 );   // n >= 2 on this path (host check)
			const f2u v = *reinterpret_cast<const f2u*>( x + a0c );
			return mk( v.x, v.y );
			}
		};
	// edge frames: pairs loaded from clamped addresses are shifted / zeroed here (AudioPV.cpp:54-62, :65)
	auto fix_raw = [&]( int t )
		{
		const int start = hop * t - W / 2;
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const int s0 = 2 * ( l + LP * q );
			const int a0 = start + s0;
			const int d = a0 - min( max( a0, 0 ), n32 - 2 );
			float v0 = ( d == 0 ) ? z[q].x : ( d == 1 ? z[q].y : 0.0f );
			float v1 = ( d == 0 ) ? z[q].y : ( d == -1 ? z[q].x : 0.0f );
			if( s0 >= W ) v0 = 0.0f;
			if( s0 + 1 >= W ) v1 = 0.0f;
			z[q] = mk( v0, v1 );
			}
		};
	auto transform_frame = [&]( int t, auto fast_tag )
		{
		if constexpr( !decltype( fast_tag )::value ) fix_raw( t );
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const cf w = s_win[LP * q];
			z[q] = mk( z[q].x * w.x, z[q].y * w.y );
			}
		fft_sub<LOG2C, LP>( z, buf, s, l );
		#pragma unroll
		for( int q = H; q < E; ++q ) buf[padl + QS * q] = z[q];
		wave_sync();
		};

	// the bins of iteration i's frame t (stored iff store_ok); requests frame tn's samples into the registers of the dying spectrum
	auto bins_of_frame = [&]( int t, int tn, int i, bool store_ok, auto halo_tag, auto next_fast )
		{
		constexpr bool halo = decltype( halo_tag )::value;
		const cf zh = buf[C / 2 + C / 16];                                    // Z[ C/2 ], slot PAD( C/2 )
		const cf z0 = z[0];                                                   // l = 0: Z[0]
		#pragma unroll
		for( int q = H; q < E; ++q ) z[q] = load_pair( tn, q, next_fast );    // the upper half is in LDS now: its registers are free
		const unsigned roff = unsigned( t - tb0 ) * unsigned( ( C + 1 ) * 8 );
		// (a lane whose frame is not to be written -- the surplus iterations of a short last chain, a spare group -- stores into the dump area: the
		// stores are never inside a branch, so that the wait for the next frame's samples can COUNT them; behind a branch it is a wait for all of them)
		cf * rowk = reinterpret_cast<cf*>( rb + ( roff + l8 ) );
		cf * rowm = reinterpret_cast<cf*>( rb + ( roff + unsigned( C * 8 ) - l8 ) );
		if( !halo && !store_ok )
			{
			rowk = reinterpret_cast<cf*>( p.dump ) + ( lane & 15 );               // (everything the stores below reach lies inside the area's 128 slots)
			rowm = reinterpret_cast<cf*>( p.dump ) + 112 + ( lane & 15 );
			}
		cf outk[halo ? 1 : H], outm[halo ? 1 : H];
		#pragma unroll
		for( int gq = 0; gq < H / NP; ++gq )
			{
			VB re, im, pv, binf, expd;
			#pragma unroll
			for( int j = 0; j < NP; ++j )
				{
				const int q = gq * NP + j;
				const cf zk = z[q];
				const cf zm = mirror[-QS * q];                                      // l = 0, q = 0 reads an unused slot: overridden below
				const cf w = s_w2[LP * q];                                          // 0.5 exp( -2 pi i k / 2C )
				const v4f_t kc = s_kc[LP * q];
				z[q] = load_pair( tn, q, next_fast );                               // Z[k] is consumed: the next frame's samples take its place
				const float sx = zk.x + zm.x, dy = zk.y + zm.y, dx = zk.x - zm.x, sy = zk.y - zm.y;
				const float t1v = __builtin_fmaf( w.x, dy, w.y * dx );
				const float t2v = __builtin_fmaf( w.x, dx, -( w.y * dy ) );
				float rk = __builtin_fmaf( 0.5f, sx, t1v ), ik = __builtin_fmaf( 0.5f, sy, -t2v );
				float rm = __builtin_fmaf( 0.5f, sx, -t1v ), imv = __builtin_fmaf( -0.5f, sy, -t2v );
				if( q == 0 )
					{
					rk = ( l == 0 ) ? z0.x + z0.y : rk;  ik = ( l == 0 ) ? 0.0f : ik;      // X[0]
					rm = ( l == 0 ) ? z0.x - z0.y : rm;  imv = ( l == 0 ) ? 0.0f : imv;    // X[C]
					}
				re[j] = rk; im[j] = ik; re[NP + j] = rm; im[NP + j] = imv;
				pv[j] = prevk[q]; pv[NP + j] = prevm[q];
				binf[j] = kc.x; expd[j] = kc.y; binf[NP + j] = kc.z; expd[NP + j] = kc.w;
				}
			VB phase, m;
			polar_v( re, im, phase, m );                                            // phase_vocoder.cpp:37-52 (AudioPV.cpp:69-73)
			if constexpr( halo )
				{
				// the halo lends its phases; a channel's first chain starts from zeros (AudioPV.cpp:44)
				const bool first = t0 == 0;
				#pragma unroll
				for( int j = 0; j < NP; ++j ) { prevk[gq * NP + j] = first ? 0.0f : phase[j]; prevm[gq * NP + j] = first ? 0.0f : phase[NP + j]; }
				}
			else
				{
				#pragma unroll
				for( int j = 0; j < NP; ++j ) { prevk[gq * NP + j] = phase[j]; prevm[gq * NP + j] = phase[NP + j]; }   // :45
				const VB phase_diff = phase - pv;                                    // :44
				const VB delta_phase = phase_diff - expd;                            // :47-48
				VB wrapped = delta_phase;
				if( use_wrapping ) wrapped = delta_phase - vsplat<VB>( FLANHIP_PI2_F ) * round_half_away_v( div_pi2_v( delta_phase ) );   // :39-42,49
				VB war;
				#pragma unroll
				for( int j = 0; j < NV; ++j ) war[j] = wrapped[j] * k_ar;
				const VB f = binf + div_pi2_v( war );                                // :50-52
				#pragma unroll
				for( int j = 0; j < NP; ++j )
					{
					outk[gq * NP + j] = cf{ m[j], f[j] };
					outm[gq * NP + j] = cf{ m[NP + j], f[NP + j] };
					}
				if constexpr( SUMS )
					{
					const VB term = div_c_each( f, k_ard ) * vsplat<VB>( FLANHIP_PI2_F );                   // phase_vocoder.cpp:57-58
					#pragma unroll
					for( int j = 0; j < NP; ++j )
						{
						sumk[gq * NP + j] += store_ok ? double( term[j] ) : 0.0;
						summ[gq * NP + j] += store_ok ? double( term[NP + j] ) : 0.0;
						}
					#pragma unroll
					for( int j = 0; j < NV; j += 2 ) mmax = store_ok ? max( mmax, max( __float_as_uint( m[j] ), __float_as_uint( m[j + 1] ) ) ) : mmax;
					}
				}
			}
		if constexpr( !halo )
			{
			// the MFs leave together at the frame's end, BEHIND every request for the next frame's samples (memory operations retire in order)
				{
				#pragma unroll
				for( int q = 0; q < H; ++q )
					{
					if constexpr( ( ABL & 2 ) != 0 ) asm volatile( "" :: "v"( outk[q].x ), "v"( outk[q].y ), "v"( outm[q].x ), "v"( outm[q].y ) );
					// 16 lanes per chain: a store instruction's four 128-byte segments all end in partial cache lines, and streamed past the L2 (non-temporal) every
					// one of them is a partial write to memory -- plain stores let the L2 put the lines together (ablation, round 6: 0.216 -> 0.167 ms; at 32 lanes
					// per chain the two forms measure the same)
					else if constexpr( ( ABL & 4 ) != 0 || LP <= 16 ) { rowk[LP * q] = outk[q]; rowm[-LP * q] = outm[q]; }
					else
						{
						__builtin_nontemporal_store( outk[q], rowk + LP * q );
						__builtin_nontemporal_store( outm[q], rowm - LP * q );
						}
					}
				}
			}
		ring = ( l == ( i & ( LP - 1 ) ) ) ? zh : ring;
		wave_sync();
		};

	// the batch of bin C/2: lane j of a group holds iteration ib + j of its chain, j < nb
	auto flush_half_bin = [&]( int ib, int nb )
		{
		const int i = ib + l;
		const int t = t0 - 1 + i;
		const bool valid = active && l < nb && i >= 1 && t < t1;
		const float re = ring.x, im = -ring.y;                                // X = conj Z[ C/2 ]
		float phase = atan2_fast( im, re );
		if( i == 0 && t0 == 0 ) phase = 0.0f;                                  // (a channel's first chain has no halo: AudioPV.cpp:44)
		float pvx = __shfl_up( phase, 1 );
		pvx = ( l == 0 ) ? prevx : pvx;
		prevx = __shfl( phase, ( lane & ~( LP - 1 ) ) + nb - 1 );
		const float bx = float( C / 2 ) * p.sample_rate * ( 1.0f / float( 2 * C ) );
		const float phase_diff = phase - pvx;
		const float delta_phase = phase_diff - div_c( bx, p.ar_div ) * FLANHIP_PI2_F;
		const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * round_half_away( div_pi2( delta_phase ) ) : delta_phase;
		const float f = bx + div_pi2( wrapped * p.analysis_rate );
		const float m = magnitude_scaled( re, im );
		if( valid && ( ABL & 3 ) == 0 ) __builtin_nontemporal_store( mk( m, f ), reinterpret_cast<cf*>( rb + ( unsigned( t - tb0 ) * unsigned( ( C + 1 ) * 8 ) + unsigned( C / 2 * 8 ) ) ) );
		if constexpr( SUMS )
			{
			const float term = valid ? div_c( f, p.ar_div ) * FLANHIP_PI2_F : 0.0f;    // phase_vocoder.cpp:57-58 (what is not a frame of the chain adds +0)
			for( int j = 0; j < nb; ++j ) sumx += double( __shfl( term, ( lane & ~( LP - 1 ) ) + j ) );   // in frame order, every lane of the group alike
			mmax = valid ? max( mmax, __float_as_uint( m ) ) : mmax;
			}
		};

	constexpr std::true_type inside{};
	constexpr std::false_type outside{};
	const int iters = p.L + 1;
	// `plain` iterations: every group's frame AND its successor lie inside the signal (the plain loads)
	auto plain = [&]( int i ) { return __all( frame_inside( frame_of( i ) ) && frame_inside( frame_of( i + 1 ) ) ) != 0; };
		{
		const int tf = frame_of( 0 );
		if( __all( frame_inside( tf ) ) )
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) z[q] = load_pair( tf, q, inside );
			transform_frame( tf, inside );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) z[q] = load_pair( tf, q, outside );
			transform_frame( tf, outside );
			}
		}
	// iteration 0, the halo
	if( plain( 0 ) ) { bins_of_frame( frame_of( 0 ), frame_of( 1 ), 0, false, std::true_type{}, inside ); transform_frame( frame_of( 1 ), inside ); }
	else { bins_of_frame( frame_of( 0 ), frame_of( 1 ), 0, false, std::true_type{}, outside ); transform_frame( frame_of( 1 ), outside ); }
	auto step = [&]( int i, auto next_fast )
		{
		const int t = t0 - 1 + i;
		const bool more = i + 1 < iters;
		bins_of_frame( frame_of( i ), frame_of( i + 1 ), i, active && t < t1, std::false_type{}, next_fast );
		if( ( ( i + 1 ) & ( LP - 1 ) ) == 0 || !more ) flush_half_bin( i - ( i & ( LP - 1 ) ), ( i & ( LP - 1 ) ) + 1 );
		if( more ) transform_frame( frame_of( i + 1 ), next_fast );
		};
	// three loops, not one with a choice inside (pv_kernels_v3.h)
	int i = 1;
	for( ; i < iters && !plain( i ); ++i ) step( i, outside );
	for( ; i < iters && plain( i ); ++i ) step( i, inside );
	for( ; i < iters; ++i ) step( i, outside );

	if constexpr( SUMS )
		{
		bool bad = mmax >= 0x7f800000u;
		auto fold = [&]( double sq ) -> double
			{
			bad |= !( __builtin_fabs( sq ) <= 1.7976931348623157e308 );              // a NaN / Inf frequency poisons its sum
			return ( __builtin_fabs( sq ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( sq ) : fold_phase_any( sq );
			};
		// ... to the workspace (what k_phase_sums2 would compute) and -- staged in this chain's now idle transform buffer -- into the block's total: one total
		// per group of NCH chains lets the synthesis kernel work out its own carries from a scan over the GROUPS (an eighth / a sixteenth of the chains; none
		// at all up to 40 groups per channel): see k_analyze_v2
		double * dst = p.sums + chain * ( C + 1 );
		double * stage = reinterpret_cast<double*>( buf );
		#pragma unroll
		for( int q = 0; q < H; ++q )
			{
			const double a = fold( sumk[q] ), b = fold( summ[q] );
			if( active ) { dst[l + LP * q] = a; dst[C - l - LP * q] = b; }
			stage[l + LP * q] = a; stage[C - l - LP * q] = b;
			}
		const double vx = fold( sumx );
		if( active && l == 0 ) dst[C / 2] = vx;
		if( l == 0 ) stage[C / 2] = vx;
		bad = bad && active;
		const bool any_bad = __any( bad );
		if( p.nan_out && lane == 0 )
			{
			if( blockIdx.x == 0 && wave == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
		if( p.group_sums )
			{
			__syncthreads();
			const int live = min( NCH, p.chains_per_channel - group * NCH );          // chains of this block that are chains of the channel
			double * gdst = p.group_sums + ( int64_t( channel ) * groups + group ) * ( C + 1 );
			for( int bin = tid; bin <= C; bin += NT )
				{
				double run = 0.0;
				for( int w = 0; w < live; ++w )
					{
					const double v = run + reinterpret_cast<const double*>( s + L::buf0( true ) + w * L::BUF_LEN )[bin];
					run = ( __builtin_fabs( v ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( v ) : fold_phase_any( v );
					}
				gdst[bin] = run;
				}
			}
		}
	}

// =================================================================================================================
// PV::convert_to_audio, dft 512 / 256; HOPQ = hop / ( 2 LP ) (the overlap-add accumulator in registers: acc[q] <-> samples pos + 2 LP q + 2 l (+1))
// =================================================================================================================
template<int LOG2C, int LP, int WAVES, int HOPQ, int OCC>
__global__ __launch_bounds__( 64 * WAVES, OCC ) void k_synthesize_sub( SynthParams p )
	{
	using L = SubLds<LOG2C, LP>;
	constexpr int C = L::C, E = 8, H = 4, G = L::G, NT = 64 * WAVES, QS = L::QS, NCH = WAVES * G, STEP = 2 * LP;
	static_assert( HOPQ == 1 || HOPQ == 2 || HOPQ == 4 || HOPQ == 8, "hop = 1, 2, 4 or 8 steps of 2 LP samples" );
	constexpr int hop = HOPQ * STEP;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 ), l = lane & ( LP - 1 ), g = lane / LP;
	const int W = p.window_size;
	const int cancel_word = cancel_peek( p.cancel );
	sub_load_twiddles<LOG2C, LP, NT>( s, p.tw, tid );
	for( int i = tid; i < C / 2; i += NT ) { const cf w = p.tw2[i]; s[L::W2 + i] = mk( w.x, -w.y ); }   // exp( +2 pi i k / 2C )
		{
		float * win = reinterpret_cast<float*>( s + L::WIN );
		for( int i = tid; i < 2 * C; i += NT ) win[i] = ( i < W ) ? p.window[i] * p.window_scale : 0.0f;      // AudioPV.cpp:102
		}
	__syncthreads();
	if( cancel_seen( cancel_word ) ) return;
	const cf * s_w2 = s + L::W2 + l;
	const cf * s_win = s + L::WIN + l;
	cf * buf = s + L::buf0( false ) + ( wave * G + g ) * L::BUF_LEN;

	const int groups = ( p.chains_per_channel + NCH - 1 ) / NCH;
	const int channel = int( blockIdx.x ) / groups, group = int( blockIdx.x ) % groups;
	const int cic_raw = group * NCH + wave * G + g;
	const bool active = cic_raw < p.chains_per_channel;
	const int cic = active ? cic_raw : p.chains_per_channel - 1;
	const int64_t chain = int64_t( channel ) * p.chains_per_channel + cic;
	const int t0 = cic * p.L;
	const int t1 = int( min( int64_t( t0 ) + p.L, p.F ) );
	const bool last_chain = cic == p.chains_per_channel - 1;
	const int nf = t1 - t0;
	float * out1 = p.out + int64_t( channel ) * p.out_len;
	cf * out2 = reinterpret_cast<cf*>( out1 );
	cf * head2 = reinterpret_cast<cf*>( p.head + chain * p.head_len );
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = cic == 0 ? INT64_MIN : chain_start + p.head_len;
	const int tb0 = group * NCH * p.L;
	const char * const pvb = reinterpret_cast<const char*>( p.pv + ( int64_t( channel ) * p.F + tb0 ) * ( C + 1 ) );
	const unsigned l8 = 8u * unsigned( l );
	const int padl = l + ( l >> 3 );
	cf * mirror = buf + ( C + C / 8 ) - l - ( ( l + 7 ) >> 3 );               // mirror[-QS q] = slot PAD( C - l - LP q )
	const DivC k_ard = p.ar_div;
	cf acc[E];
	#pragma unroll
	for( int q = 0; q < E; ++q ) acc[q] = mk( 0.0f, 0.0f );
	cf * dump2 = reinterpret_cast<cf*>( p.dump ) + lane;

	// one step of 2 LP samples leaves the chain: one store per step, never inside a branch (what is not to be written goes to the dump area)
	auto emit_step = [&]( int64_t a0, cf v, bool ok )
		{
		const int64_t a = a0 + 2 * l;
		cf * dst = ( a0 < own_start ) ? head2 + ( ( a - chain_start ) >> 1 ) : out2 + ( a >> 1 );
		if( !ok || ( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) ) dst = dump2;
		*dst = v;
		};
	cf mfk[H], mfm[H], mfx;
	auto load_row = [&]( int t )                                              // t: clamped into the chain
		{
		const unsigned ro = unsigned( t - tb0 ) * unsigned( ( C + 1 ) * 8 );
		const cf * row = reinterpret_cast<const cf*>( pvb + ro );
		const cf * rowk = reinterpret_cast<const cf*>( pvb + ( ro + l8 ) );
		const cf * rowm = reinterpret_cast<const cf*>( pvb + ( ro + unsigned( C * 8 ) - l8 ) );
		#pragma unroll
		for( int q = 0; q < H; ++q )
			{
			mfk[q] = __builtin_nontemporal_load( rowk + LP * q );
			mfm[q] = __builtin_nontemporal_load( rowm - LP * q );
			}
		mfx = __builtin_nontemporal_load( row + C / 2 );
		};
	// phase_buffer (AudioPV.cpp:105) on entry to the chain
	double phk[H], phm[H], phx;
	if( p.group_sums )
		{
		// No scan over the chains ran: `carry` still holds the chains' own sums.  The running phase on entry to a chain = the carry of its group (group_carry,
		// from a scan over the group totals, or the totals of the groups before this one added up here) + the chains of this group before it, added and folded
		// in order -- k_synthesize_v2's carry prologue with NCH chains per block: one thread per bin, every load ahead of the dependent additions; every chain's
		// carries land in its (still idle) transform buffer
		auto fold = []( double r ) { return ( __builtin_fabs( r ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( r ) : fold_phase_any( r ); };
		const double * gs = ( p.group_carry ? p.group_carry : p.group_sums ) + int64_t( channel ) * groups * ( C + 1 );
		const double * sums0 = p.carry + ( int64_t( channel ) * p.chains_per_channel + int64_t( group ) * NCH ) * ( C + 1 );
		const int live = min( NCH, p.chains_per_channel - group * NCH );
		constexpr int NB = ( C + NT ) / NT;                                       // bins per thread
		int bins_of[NB]; bool has[NB]; double run[NB];
		#pragma unroll
		for( int b = 0; b < NB; ++b ) { bins_of[b] = tid + NT * b; has[b] = bins_of[b] <= C; if( !has[b] ) bins_of[b] = C; run[b] = 0.0; }
		double vc[NB][NCH];
		#pragma unroll
		for( int b = 0; b < NB; ++b )
			{
			#pragma unroll
			for( int w = 0; w < NCH; ++w ) vc[b][w] = ( w < live ) ? sums0[int64_t( w ) * ( C + 1 ) + bins_of[b]] : 0.0;
			}
		if( p.group_carry )
			{
			#pragma unroll
			for( int b = 0; b < NB; ++b ) run[b] = gs[int64_t( group ) * ( C + 1 ) + bins_of[b]];
			}
		else
			{
			for( int g0 = 0; g0 < group; g0 += 16 )
				{
				double v[NB][16];
				#pragma unroll
				for( int b = 0; b < NB; ++b )
					{
					#pragma unroll
					for( int u = 0; u < 16; ++u ) v[b][u] = ( g0 + u < group ) ? gs[int64_t( g0 + u ) * ( C + 1 ) + bins_of[b]] : 0.0;
					}
				#pragma unroll
				for( int u = 0; u < 16; ++u )
					{
					#pragma unroll
					for( int b = 0; b < NB; ++b ) run[b] = fold( run[b] + v[b][u] );       // + 0.0 past the end: fold( x ) of a folded x is x
					}
				}
			}
		#pragma unroll
		for( int w = 0; w < NCH; ++w )
			{
			#pragma unroll
			for( int b = 0; b < NB; ++b )
				{
				if( has[b] ) reinterpret_cast<double*>( s + L::buf0( false ) + w * L::BUF_LEN )[bins_of[b]] = run[b];
				run[b] = fold( run[b] + vc[b][w] );
				}
			}
		if( tid == 0 && blockIdx.x == 0 )
			{
			if( p.nan_in && p.nan_flag && p.nan_in[0] == p.nan_in[2] && p.nan_in[2] != 0 ) atomicOr( p.nan_flag, 1 );
			if( p.expect_epoch && p.nan_in && p.nan_flag && p.nan_in[2] != p.expect_epoch ) atomicOr( p.nan_flag, 2 );   // the sums in this workspace are not the noted producer's
			if( p.skip_words ) const_cast<int*>( p.skip_words )[4] = 0;              // a handed-over pre-pass is good for one convert_to_audio
			}
		__syncthreads();
		const double * mine = reinterpret_cast<const double*>( buf );
		#pragma unroll
		for( int q = 0; q < H; ++q ) { phk[q] = mine[l + LP * q]; phm[q] = mine[C - l - LP * q]; }
		phx = mine[C / 2];
		wave_sync();
		}
	else
		{
		const double * carry = p.carry + chain * ( C + 1 );
		#pragma unroll
		for( int q = 0; q < H; ++q ) { phk[q] = carry[l + LP * q]; phm[q] = carry[C - l - LP * q]; }
		phx = carry[C / 2];
		}
	cf z[E];
	auto bins_of_row = [&]()
		{
		bool slow = false;
		float dk[H], dm[H], dx;
		if( k_ard.exact )
			{
			auto div_exact = [&]( float x ) { const float q0 = x * k_ard.rc; return __builtin_fmaf( __builtin_fmaf( -q0, k_ard.c, x ), k_ard.rc, q0 ); };   // pv_math.h: div_c
			#pragma unroll
			for( int q = 0; q < H; ++q ) { dk[q] = div_exact( mfk[q].y ); dm[q] = div_exact( mfm[q].y ); }
			dx = div_exact( mfx.y );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < H; ++q ) { dk[q] = mfk[q].y / k_ard.c; dm[q] = mfm[q].y / k_ard.c; }
			dx = mfx.y / k_ard.c;
			}
		#pragma unroll
		for( int q = 0; q < H; ++q )
			{
			phk[q] += double( dk[q] * FLANHIP_PI2_F );                             // phase_vocoder.cpp:57-58
			phm[q] += double( dm[q] * FLANHIP_PI2_F );
			slow |= !( __builtin_fabs( phk[q] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) ) || !( __builtin_fabs( phm[q] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
			}
		phx += double( dx * FLANHIP_PI2_F );
		slow |= !( __builtin_fabs( phx ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
		cf xk[H], xm[H], xx;
		if( __any( slow ) )
			{
			#pragma unroll
			for( int q = 0; q < H; ++q )
				{
				phk[q] = fold_phase_any( phk[q] ); phm[q] = fold_phase_any( phm[q] );
				const float2 a = sincos_wide( float( phk[q] ) ), b = sincos_wide( float( phm[q] ) );
				xk[q] = mk( mfk[q].x * a.y, mfk[q].x * a.x );
				xm[q] = mk( mfm[q].x * b.y, mfm[q].x * b.x );
				}
			phx = fold_phase_any( phx );
			const float2 a = sincos_wide( float( phx ) );
			xx = mk( mfx.x * a.y, mfx.x * a.x );
			}
		else
			{
			#pragma unroll
			for( int q0 = 0; q0 < H; q0 += 2 )
				{
				v4f th, m4;
				#pragma unroll
				for( int i = 0; i < 2; ++i )
					{
					phk[q0 + i] = fold_phase_loop( phk[q0 + i] );                  // phase_vocoder.cpp:59
					phm[q0 + i] = fold_phase_loop( phm[q0 + i] );
					th[i] = float( phk[q0 + i] ); th[2 + i] = float( phm[q0 + i] );
					m4[i] = mfk[q0 + i].x; m4[2 + i] = mfm[q0 + i].x;
					}
				v4f sn, cs;
				sincos_fast_v( th, sn, cs );
				const v4f xr = m4 * cs, xi = m4 * sn;                            // std::polar, :60
				#pragma unroll
				for( int i = 0; i < 2; ++i ) { xk[q0 + i] = cf{ xr[i], xi[i] }; xm[q0 + i] = cf{ xr[2 + i], xi[2 + i] }; }
				}
			phx = fold_phase_loop( phx );
			float sn, cs;
			sincos_fast( float( phx ), sn, cs );
			xx = mk( mfx.x * cs, mfx.x * sn );
			}
		// ---- merge (see k_synthesize_v2)
		#pragma unroll
		for( int q = 0; q < H; ++q )
			{
			cf a = xk[q], b = xm[q];                                            // X[k], X[C-k]
			if( q == 0 ) { a.y = ( l == 0 ) ? 0.0f : a.y; b.y = ( l == 0 ) ? 0.0f : b.y; }   // c2r ignores Im X[0], Im X[C]
			const cf w = s_w2[LP * q];
			const float ax = a.x + b.x, ay = a.y - b.y;
			const float dx2 = a.x - b.x, dy = a.y + b.y;
			const float bx = __builtin_fmaf( w.x, dx2, -( w.y * dy ) ), by = __builtin_fmaf( w.x, dy, w.y * dx2 );
			z[q] = mk( ax - by, -( ay + bx ) );                                 // Z[k]
			mirror[-QS * q] = mk( ax + by, ay - bx );                           // Z[C-k] (l = 0, q = 0: an unused slot)
			}
		if( l == 0 ) buf[C / 2 + C / 16] = mk( 2.0f * xx.x, 2.0f * xx.y );      // Z[C/2] = 2 X[C/2]
		wave_sync();
		#pragma unroll
		for( int q = H; q < E; ++q ) z[q] = buf[padl + QS * q];
		wave_sync();
		};

	auto frame_of = [&]( int i ) { return min( t0 + i, t1 - 1 ); };
	load_row( frame_of( 0 ) );
	bins_of_row();
	int64_t pos = chain_start;
	for( int i = 0; i < p.L; ++i )
		{
		const bool ok = active && i < nf;
		if( i + 1 < p.L ) load_row( frame_of( i + 1 ) );
		fft_sub<LOG2C, LP>( z, buf, s, l );
		// G = fft( conj Z ): x[2n] = G[n].x, x[2n+1] = -G[n].y; window and overlap-add (AudioPV.cpp:122-134)
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const cf w = s_win[LP * q];                                         // zero beyond W
			acc[q].x += ok ? z[q].x * w.x : 0.0f;
			acc[q].y += ok ? ( -z[q].y ) * w.y : 0.0f;
			}
		#pragma unroll
		for( int q = 0; q < HOPQ; ++q ) emit_step( pos + STEP * q, acc[q], ok );
		if( ok )
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) acc[q] = ( q + HOPQ < E ) ? acc[q + HOPQ] : mk( 0.0f, 0.0f );
			pos += hop;
			}
		if( i + 1 < p.L ) bins_of_row();
		}
	// flush the partial sums that the next chain's head completes; the last chain zero-fills to the end of the output
	const int64_t ring_end = pos + ( W - hop );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	#pragma unroll
	for( int q = 0; q < E; ++q )
		{
		const int64_t a0 = pos + STEP * q;
		emit_step( a0, acc[q], active && a0 < flush_end );
		}
	if( active ) for( int64_t a0 = pos + STEP * E; a0 < flush_end; a0 += STEP ) emit_step( a0, mk( 0.0f, 0.0f ), true );
	}

} // namespace flanhip
