// pv_kernels_v3.h -- the dft 1024 and dft 512 analysis / synthesis kernels (C = 512 / 256 complex points per frame), round 5.
//
// pv_kernels_v2.h's design at the two sizes below it: ONE wavefront per chain, the frame in registers, bin PAIRS ( k, C - k ) per lane, the
// rotated frame loop with one counted wait, scalar base + 32-bit offsets.  What differs is the transform and the occupancy:
//   * a lane holds E = C / 64 complex points (8 / 4) and every pass is ONE radix-E butterfly per lane: C = 64 E = E^P, P = 3 (8 x 8 x 8)
//     or 4 (4 x 4 x 4 x 4) Stockham passes with P - 1 transposes through the wavefront's own LDS buffer; element i sits at slot
//     i + i / E (one pad slot per butterfly's worth: the scatter writes of every pass land on distinct banks);
//   * the state of a chain is 4 (2) pairs + bin C/2 instead of 8 + 1: ~128 (~64) registers, so a SIMD holds four (eight) wavefronts where the
//     dft 2048 kernels hold two -- 16 (32) chains per CU;
//   * twiddles of all passes come from ONE table [pass][r][k] built in the prologue from the plan's exp( -2 pi i k / C ).
// Reference: Conversions/AudioPV.cpp:12-78 / :86-139, phase_vocoder.cpp:37-61.
#pragma once
#include "pv_kernels_v2.h"

namespace flanhip {

template<int LOG2C> struct V3Lds
	{
	static_assert( LOG2C == 9 || LOG2C == 8, "dft 1024 / 512" );
	static constexpr int C = 1 << LOG2C;
	static constexpr int E = C / 64;                         // points per lane = the radix
	static constexpr int P = ( LOG2C == 9 ) ? 3 : 4;         // passes: E^P = C
	static constexpr int QS = 64 + 64 / E;                   // slots between elements i and i + 64 (one lane's consecutive points)
	static constexpr int ipow( int b, int e ) { return e == 0 ? 1 : b * ipow( b, e - 1 ); }
	static constexpr int tw_off( int pass ) { return pass <= 1 ? 0 : tw_off( pass - 1 ) + ( E - 1 ) * ipow( E, pass - 1 ); }   // table of pass 1 .. P - 1: [E - 1][NS], NS = E^pass
	static constexpr int TW_LEN = tw_off( P );
	static constexpr int TW = 0;
	static constexpr int W2 = TW + TW_LEN;                   // [C/2]  exp( -+ 2 pi i k / 2C ) (analysis: halved)
	static constexpr int WIN = W2 + C / 2;                   // [2C floats]
	static constexpr int KC = WIN + C;                       // analysis: [C/2] float4 { binf(k), expected(k), binf(C-k), expected(C-k) }
	static constexpr int BUF_LEN = C + C / E + 1;            // highest slot: PAD( C ); also holds C + 1 doubles (the chain's sums / carries)
	static constexpr int buf0( bool analysis ) { return analysis ? KC + C : KC; }
	static constexpr size_t bytes( int waves, bool analysis ) { return size_t( buf0( analysis ) + waves * BUF_LEN ) * 8; }
	static_assert( TW_LEN % 2 == 0, "the float4 table is 16-byte aligned" );
	};

// the P passes on the register array (natural layout in and out: z[q] = element lane + 64 q); s_tw: the table above
template<int LOG2C, int PASS>
__device__ __forceinline__ void fft_v3_pass( cf ( &z )[V3Lds<LOG2C>::E], cf * buf, const cf * s_tw, int lane )
	{
	using L = V3Lds<LOG2C>;
	constexpr int R = L::E, NS = L::ipow( R, PASS );
	// butterfly j = lane: inputs in[ j + 64 r ], k = j mod NS, twiddles exp( -2 pi i r k / ( NS R ) ), outputs out[ ( j - k ) R + k + NS r ]
	const int padl = lane + lane / R;
	const cf * rp = buf + padl;
	cf v[R], tw[R - 1];
	#pragma unroll
	for( int r = 0; r < R; ++r ) v[r] = rp[L::QS * r];
	const int k = lane & ( NS - 1 );
	const cf * tp = s_tw + L::tw_off( PASS ) + k;
	#pragma unroll
	for( int r = 1; r < R; ++r ) tw[r - 1] = tp[( r - 1 ) * NS];          // requested with the points, ahead of the fence: one LDS round trip
	wave_sync();
	#pragma unroll
	for( int r = 1; r < R; ++r ) v[r] = cmul( v[r], tw[r - 1] );
	dft_reg<R>( v );
	if constexpr( PASS < L::P - 1 )
		{
		const int hi = lane - k;
		cf * wp = buf + hi * ( R + 1 ) + k + k / R;
		#pragma unroll
		for( int r = 0; r < R; ++r ) wp[( NS + NS / R ) * r] = v[r];
		wave_sync();
		fft_v3_pass<LOG2C, PASS + 1>( z, buf, s_tw, lane );
		}
	else
		{
		#pragma unroll
		for( int r = 0; r < R; ++r ) z[r] = v[r];                            // NS = 64: out[ lane + 64 r ], the natural register layout
		wave_sync();
		}
	}
template<int LOG2C>
__device__ __forceinline__ void fft_v3( cf ( &z )[V3Lds<LOG2C>::E], cf * buf, const cf * s_tw, int lane )
	{
	constexpr int R = V3Lds<LOG2C>::E;
	dft_reg<R>( z );                                                         // pass 0: inputs lane + 64 r, outputs out[ R lane + r ]
	cf * wp = buf + ( R + 1 ) * lane;
	#pragma unroll
	for( int r = 0; r < R; ++r ) wp[r] = z[r];
	wave_sync();
	fft_v3_pass<LOG2C, 1>( z, buf, s_tw, lane );
	}

template<int LOG2C, int NT> __device__ __forceinline__ void v3_load_twiddles( cf * s_tw, const cf * tw, int tid )
	{
	using L = V3Lds<LOG2C>;
	constexpr int R = L::E;
	int off = 0, ns = R;
	#pragma unroll
	for( int pass = 1; pass < L::P; ++pass )
		{
		for( int i = tid; i < ( R - 1 ) * ns; i += NT )
			{
			const int r = i / ns + 1, k = i % ns;
			s_tw[off + i] = tw[r * k * ( L::C / ( ns * R ) )];                  // exp( -2 pi i r k / ( NS R ) )
			}
		off += ( R - 1 ) * ns;
		ns *= R;
		}
	}

// =================================================================================================================
// Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78), dft 1024 / 512.  See k_analyze_v2 for the reasons behind the loop's shape.
// =================================================================================================================
// PF: frames the sample requests run ahead of their use (1: like k_analyze_v2, into the registers of the dying spectrum; 2: a frame further, through
// a second register set -- the requests a transform waits for then sit IN FRONT of the last frame's MF stores in issue order, and the stores in front of
// them are two frames old: at these sizes a frame is too short for one frame's lead to cover the acknowledgement of a row of stores)
template<int LOG2C, int WAVES, bool SUMS, int OCC, int NV = 2 * ( V3Lds<LOG2C>::E / 2 ), int PF = 1, int ABL = 0>
__global__ __launch_bounds__( 64 * WAVES, OCC ) void k_analyze_v3( AnalyzeParams p )
	{
	using L = V3Lds<LOG2C>;
	constexpr int C = L::C, E = L::E, H = E / 2, NT = 64 * WAVES, QS = L::QS;
	constexpr int NP = NV / 2;                                                  // bin pairs evaluated together as one vector stream
	static_assert( NV >= 2 && H % NP == 0, "whole groups of pairs" );
	using VB = FA<NV>;                                                          // N separate floats, not a register tuple (pv_math.h)
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 );   // (a scalar: frame ranges, loop counts and base addresses live in SGPRs, the loops are scalar loops)
	const int W = p.window_size, hop = p.hop;
	const int cancel_word = cancel_peek( p.cancel );

	// ---- tables (block-wide) --------------------------------------------------------------------------------------------
	v3_load_twiddles<LOG2C, NT>( s + L::TW, p.tw, tid );
	for( int i = tid; i < C / 2; i += NT ) { const cf w = p.tw2[i]; s[L::W2 + i] = mk( 0.5f * w.x, 0.5f * w.y ); }
		{
		float * win = reinterpret_cast<float*>( s + L::WIN );
		for( int i = tid; i < 2 * C; i += NT ) win[i] = ( i < W ) ? p.window[i] : 0.0f;          // AudioPV.cpp:60,65
		// bin frequency (PVBuffer.cpp:443-446: the division by dft, a power of two, is exactly a multiplication) and expected phase
		// advance (phase_vocoder.cpp:47) of the pair ( k, C - k )
		v4f_t * kc = reinterpret_cast<v4f_t*>( s + L::KC );
		const float rdft = 1.0f / float( 2 * C );
		for( int k = tid; k < C / 2; k += NT )
			{
			const float bk = float( k ) * p.sample_rate * rdft, bm = float( C - k ) * p.sample_rate * rdft;
			kc[k] = v4f_t{ bk, div_c( bk, p.ar_div ) * FLANHIP_PI2_F, bm, div_c( bm, p.ar_div ) * FLANHIP_PI2_F };
			}
		}
	__syncthreads();
	const cf * s_tw = s + L::TW;
	const cf * s_w2 = s + L::W2 + lane;
	const cf * s_win = s + L::WIN + lane;
	const v4f_t * s_kc = reinterpret_cast<const v4f_t*>( s + L::KC ) + lane;
	cf * buf = s + L::buf0( true ) + wave * L::BUF_LEN;
	// a block is a GROUP: WAVES consecutive chains of ONE channel (the last group of a channel may be short: its spare wavefronts idle)
	const int groups = ( p.chains_per_channel + WAVES - 1 ) / WAVES;
	const int channel = int( blockIdx.x ) / groups, group = int( blockIdx.x ) % groups;
	const int chain_in_channel = group * WAVES + wave;
	const bool active = chain_in_channel < p.chains_per_channel && !cancel_seen( cancel_word );
	const int64_t chain = int64_t( channel ) * p.chains_per_channel + ( active ? chain_in_channel : 0 );
	const int t0 = ( active ? chain_in_channel : 0 ) * p.L;
	const int t1 = int( min( int64_t( t0 ) + p.L, p.F ) );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const int n32 = int( p.n );
	// addresses: a scalar base per block + a 32-bit per-lane byte offset (everything a block touches lies within WAVES L + 1 frames of tb0)
	const int tb0 = group * WAVES * p.L - 1;
	const char * const xb = reinterpret_cast<const char*>( x ) + ( int64_t( hop ) * tb0 - W / 2 ) * 4;
	char * const rb = reinterpret_cast<char*>( p.out + ( int64_t( channel ) * p.F + tb0 ) * ( C + 1 ) );
	const unsigned lane8 = 8u * unsigned( lane );
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37
	const float k_ar = p.analysis_rate;
	const DivC k_ard = p.ar_div;
	const int padl = lane + lane / E;
	const cf * mirror = buf + ( C - 64 ) / E * ( E + 1 ) + ( 64 - lane ) + ( 64 - lane ) / E;   // mirror[-QS q] = slot PAD( C - lane - 64 q )
	struct __attribute__(( packed, aligned( 4 ) )) f2u { float x, y; };      // a sample pair at any 4-byte aligned address

	// state that crosses frames: previous phases (phase_vocoder.cpp:45) of the lane's pairs and of bin C/2
	float prevk[H], prevm[H], prevx = 0.0f;
	#pragma unroll
	for( int q = 0; q < H; ++q ) { prevk[q] = 0.0f; prevm[q] = 0.0f; }        // AudioPV.cpp:44
	double sumk[SUMS ? H : 1], summ[SUMS ? H : 1], sumx = 0.0;                // fused round trip: the chain's phase increments (phase_vocoder.cpp:57-58)
	if constexpr( SUMS )
		{
		#pragma unroll
		for( int q = 0; q < H; ++q ) { sumk[q] = 0.0; summ[q] = 0.0; }
		}
	unsigned mmax = 0u;                                                       // maximum of the magnitudes' bit patterns (Inf / NaN scan)
	cf ring = mk( 0.0f, 0.0f );                                               // Z[ C/2 ] of the chain's i-th frame waits in lane i % 64 (see k_analyze_v2)

	const int tfirst = t0 > 0 ? t0 - 1 : t0;
	const bool w_whole = ( W & 127 ) == 0;
	auto frame_inside = [&]( int t ) { return w_whole && hop * t - W / 2 >= 0 && hop * t - W / 2 + 2 * C <= n32; };

	cf z[E], raw[PF == 2 ? E : 1];
	auto run_chain = [&]()
		{
		auto load_pair = [&]( int t, int q, auto fast_tag ) -> cf
			{
			constexpr bool FAST = decltype( fast_tag )::value;
			if constexpr( ( ABL & 16 ) != 0 ) return mk( float( t ) * 1e-9f + 0.25f, float( q ) );      // (timing only: no sample loads)
			else if constexpr( FAST )
				{
				const unsigned off = unsigned( hop * ( t - tb0 ) ) * 4u + lane8;
				const f2u v = *reinterpret_cast<const f2u*>( xb + off + 512 * q );
				return mk( v.x, v.y );
				}
			else
				{
				const int start = hop * t - W / 2;
				const int a0c = min( max( start + 2 * ( lane + 64 * q ), 0 ), n32 - 2 );   // n >= 2 on this path (host check)
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
				const int s0 = 2 * ( lane + 64 * q );
				const int a0 = start + s0;
				const int d = a0 - min( max( a0, 0 ), n32 - 2 );                     // 0: pair loaded as is; -1 / +1: shifted by one; else outside
				float v0 = ( d == 0 ) ? z[q].x : ( d == 1 ? z[q].y : 0.0f );
				float v1 = ( d == 0 ) ? z[q].y : ( d == -1 ? z[q].x : 0.0f );
				if( s0 >= W ) v0 = 0.0f;
				if( s0 + 1 >= W ) v1 = 0.0f;
				z[q] = mk( v0, v1 );
				}
			};
		// window (AudioPV.cpp:60), transform; leaves the upper half of Z in buf (natural order) for the mirror reads
		auto transform_frame = [&]( int t, auto fast_tag )
			{
			if constexpr( !decltype( fast_tag )::value ) fix_raw( t );
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				const cf w = s_win[64 * q];
				z[q] = mk( z[q].x * w.x, z[q].y * w.y );
				}
			fft_v3<LOG2C>( z, buf, s_tw, lane );
			#pragma unroll
			for( int q = H; q < E; ++q ) buf[padl + QS * q] = z[q];
			wave_sync();
			};

		// the rotated loop body: [ per-bin work of frame t, which also requests frame tn's samples and stores frame t's MFs ]; the caller
		// then waits for those samples and transforms frame tn.  HALO: frame t0 - 1, of which only the phases are wanted.
		// z[q] <- pair q of the frame after this one; tn: the frame REQUESTED now (PF = 1: that same frame; PF = 2: the one after it, and the
		// frame handed to z arrived in `raw` a frame ago)
		auto next_pair = [&]( int tn, int q, auto fast_tag )
			{
			if constexpr( PF == 2 ) { z[q] = raw[q]; raw[q] = load_pair( tn, q, fast_tag ); }
			else z[q] = load_pair( tn, q, fast_tag );
			};
		auto bins_of_frame = [&]( int t, int tn, int fi, auto halo_tag, auto next_fast )
			{
			constexpr bool halo = decltype( halo_tag )::value;
			const cf zh = buf[C / 2 + C / 2 / E];                                 // Z[ C/2 ], slot PAD( C/2 )
			const cf z0 = z[0];                                                   // lane 0: Z[0]
			#pragma unroll
			for( int q = H; q < E; ++q ) next_pair( tn, q, next_fast );           // the upper half is in LDS now: its registers are free
			const unsigned roff = unsigned( t - tb0 ) * unsigned( ( C + 1 ) * 8 );
			cf * rowk = reinterpret_cast<cf*>( rb + ( roff + lane8 ) );
			cf * rowm = reinterpret_cast<cf*>( rb + ( roff + unsigned( C * 8 ) - lane8 ) );
			cf outk[halo ? 1 : H], outm[halo ? 1 : H];
			#pragma unroll
			for( int g = 0; g < H / NP; ++g )
				{
				VB re, im, pv, binf, expd;
				#pragma unroll
				for( int i = 0; i < NP; ++i )
					{
					const int q = g * NP + i;
					// bins k = lane + 64 q and C - k of the real transform from Z[k] (own) and Z[C-k] (mirror lane, through LDS)
					const cf zk = z[q];
					const cf zm = mirror[-QS * q];                                  // lane 0, q = 0 reads an unused slot: overridden below
					const cf w = s_w2[64 * q];                                      // 0.5 exp( -2 pi i k / 2C )
					const v4f_t kc = s_kc[64 * q];
					next_pair( tn, q, next_fast );                                  // Z[k] is consumed: next frame's samples take its place
					const float sx = zk.x + zm.x, dy = zk.y + zm.y, dx = zk.x - zm.x, sy = zk.y - zm.y;
					const float t1v = __builtin_fmaf( w.x, dy, w.y * dx );
					const float t2v = __builtin_fmaf( w.x, dx, -( w.y * dy ) );
					float rk = __builtin_fmaf( 0.5f, sx, t1v ), ik = __builtin_fmaf( 0.5f, sy, -t2v );
					float rm = __builtin_fmaf( 0.5f, sx, -t1v ), imv = __builtin_fmaf( -0.5f, sy, -t2v );
					if( q == 0 )
						{
						rk = ( lane == 0 ) ? z0.x + z0.y : rk;  ik = ( lane == 0 ) ? 0.0f : ik;      // X[0]
						rm = ( lane == 0 ) ? z0.x - z0.y : rm;  imv = ( lane == 0 ) ? 0.0f : imv;    // X[C]
						}
					re[i] = rk; im[i] = ik; re[NP + i] = rm; im[NP + i] = imv;
					pv[i] = prevk[q]; pv[NP + i] = prevm[q];
					binf[i] = kc.x; expd[i] = kc.y; binf[NP + i] = kc.z; expd[NP + i] = kc.w;
					}
				// phase_vocoder.cpp:37-52 (AudioPV.cpp:69-73)
				VB phase, m;
				polar_v( re, im, phase, m );
				#pragma unroll
				for( int i = 0; i < NP; ++i ) { prevk[g * NP + i] = phase[i]; prevm[g * NP + i] = phase[NP + i]; }   // :45
				if constexpr( !halo )
					{
					const VB phase_diff = phase - pv;                                // == float( double(phase) - double(prev) ), :44
					const VB delta_phase = phase_diff - expd;                        // :47-48
					VB wrapped = delta_phase;
					if( use_wrapping ) wrapped = delta_phase - vsplat<VB>( FLANHIP_PI2_F ) * round_half_away_v( div_pi2_v( delta_phase ) );   // :39-42,49
					VB war;
					#pragma unroll
					for( int i = 0; i < NV; ++i ) war[i] = wrapped[i] * k_ar;
					const VB f = binf + div_pi2_v( war );                            // :50-52
					#pragma unroll
					for( int i = 0; i < NP; ++i )
						{
						outk[g * NP + i] = cf{ m[i], f[i] };
						outm[g * NP + i] = cf{ m[NP + i], f[NP + i] };
						}
					if constexpr( SUMS )
						{
						const VB term = div_c_each( f, k_ard ) * vsplat<VB>( FLANHIP_PI2_F );                   // phase_vocoder.cpp:57-58
						#pragma unroll
						for( int i = 0; i < NP; ++i )
							{
							sumk[g * NP + i] += double( term[i] );
							summ[g * NP + i] += double( term[NP + i] );
							}
						#pragma unroll
						for( int i = 0; i < NV; i += 2 ) mmax = max( mmax, max( __float_as_uint( m[i] ), __float_as_uint( m[i + 1] ) ) );   // v_max3_u32
						}
					}
				}
			if constexpr( !halo )
				{
				// the MFs leave together at the frame's end, BEHIND every request for the next frame's samples (memory operations retire in order)
				#pragma unroll
				for( int q = 0; q < H; ++q )
					{
					if constexpr( ( ABL & 8 ) != 0 ) asm volatile( "" :: "v"( outk[q].x ), "v"( outk[q].y ), "v"( outm[q].x ), "v"( outm[q].y ) );   // (timing only: no MF stores)
					else if constexpr( ( ABL & 32 ) != 0 ) { rowk[64 * q] = outk[q]; rowm[-64 * q] = outm[q]; }                                  // (A/B: plain stores)
					else
						{
						__builtin_nontemporal_store( outk[q], rowk + 64 * q );
						__builtin_nontemporal_store( outm[q], rowm - 64 * q );
						}
					}
				}
			ring = ( lane == ( fi & 63 ) ) ? zh : ring;
			wave_sync();
			};

		// the batch of bin C/2: lane j holds frame tb + j, j < nb (frame tfirst, the halo, only lends its phase)
		auto flush_half_bin = [&]( int tb, int nb )
			{
			const int t = tb + lane;
			const bool valid = lane < nb && t >= t0;
			const float re = ring.x, im = -ring.y;                                // X = conj Z[ C/2 ]
			const float phase = atan2_fast( im, re );
			float pvx = __shfl_up( phase, 1 );
			pvx = ( lane == 0 ) ? prevx : pvx;
			prevx = __shfl( phase, nb - 1 );
			const float bx = float( C / 2 ) * p.sample_rate * ( 1.0f / float( 2 * C ) );
			const float phase_diff = phase - pvx;
			const float delta_phase = phase_diff - div_c( bx, p.ar_div ) * FLANHIP_PI2_F;
			const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * round_half_away( div_pi2( delta_phase ) ) : delta_phase;
			const float f = bx + div_pi2( wrapped * p.analysis_rate );
			const float m = magnitude_scaled( re, im );
			if( valid ) __builtin_nontemporal_store( mk( m, f ), reinterpret_cast<cf*>( rb + ( unsigned( t - tb0 ) * unsigned( ( C + 1 ) * 8 ) + unsigned( C / 2 * 8 ) ) ) );
			if constexpr( SUMS )
				{
				const float term = div_c( f, p.ar_div ) * FLANHIP_PI2_F;             // phase_vocoder.cpp:57-58
				const int j0 = __builtin_amdgcn_readfirstlane( ( tb < t0 ) ? 1 : 0 ), j1 = __builtin_amdgcn_readfirstlane( nb );
				for( int j = j0; j < j1; ++j ) sumx += double( __uint_as_float( __builtin_amdgcn_readlane( __float_as_uint( term ), j ) ) );   // in frame order
				mmax = valid ? max( mmax, __float_as_uint( m ) ) : mmax;
				}
			};

		constexpr std::true_type inside{};
		constexpr std::false_type outside{};
		// frame t's step requests frame min( t + PF, t1 - 1 ) (the last frames request the last one again: nobody waits for it) and transforms
		// frame t + 1; `inside` bodies need both of them inside the signal
		auto req = [&]( int t ) { return min( t + PF, t1 - 1 ); };
		auto plain = [&]( int t ) { return frame_inside( min( t + 1, t1 - 1 ) ) && frame_inside( req( t ) ); };
		if( frame_inside( tfirst ) && ( PF == 1 || frame_inside( req( tfirst - 1 ) ) ) )
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) z[q] = load_pair( tfirst, q, inside );
			if constexpr( PF == 2 ) { _Pragma( "unroll" ) for( int q = 0; q < E; ++q ) raw[q] = load_pair( req( tfirst - 1 ), q, inside ); }
			transform_frame( tfirst, inside );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) z[q] = load_pair( tfirst, q, outside );
			if constexpr( PF == 2 ) { _Pragma( "unroll" ) for( int q = 0; q < E; ++q ) raw[q] = load_pair( req( tfirst - 1 ), q, outside ); }
			transform_frame( tfirst, outside );
			}
		int fi = 0;
		if( t0 > 0 )
			{
			if( plain( t0 - 1 ) ) { bins_of_frame( t0 - 1, req( t0 - 1 ), fi, std::true_type{}, inside ); transform_frame( t0, inside ); }
			else { bins_of_frame( t0 - 1, req( t0 - 1 ), fi, std::true_type{}, outside ); transform_frame( t0, outside ); }
			++fi;
			}
		// Three loops, not one with a choice inside: the frames whose SUCCESSORS reach outside the signal (a few at either end of a channel), and
		// between them the loop every other frame takes, which holds the plain loads only -- with both bodies in one loop the edge body's
		// invariants (clamped addresses, window masks per point) are hoisted in front of it and spill under the register cap.
		auto step = [&]( int t, auto next_fast )
			{
			const bool more = t + 1 < t1;
			bins_of_frame( t, req( t ), fi, std::false_type{}, next_fast );
			++fi;
			if( ( fi & 63 ) == 0 || !more ) flush_half_bin( t + 1 - ( ( ( fi - 1 ) & 63 ) + 1 ), ( ( fi - 1 ) & 63 ) + 1 );
			if( more ) transform_frame( t + 1, next_fast );
			};
		int t = t0;
		for( ; t < t1 && !plain( t ); ++t ) step( t, outside );
		for( ; t < t1 && plain( t ); ++t ) step( t, inside );
		for( ; t < t1; ++t ) step( t, outside );
		};
	if( active ) run_chain();

	if constexpr( SUMS )
		{
		// the chain's sums, folded like phase_vocoder.cpp:59, go to the workspace (what k_phase_sums2 would compute) and -- staged in this
		// wavefront's now idle transform buffer -- into the group's total (see k_analyze_v2)
		double * stage = reinterpret_cast<double*>( buf );
		bool bad = mmax >= 0x7f800000u;
		auto fold = [&]( double sq ) -> double
			{
			bad |= !( __builtin_fabs( sq ) <= 1.7976931348623157e308 );              // a NaN / Inf frequency poisons its sum
			return ( __builtin_fabs( sq ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( sq ) : fold_phase_any( sq );
			};
		if( active )
			{
			double * dst = p.sums + chain * ( C + 1 );
			#pragma unroll
			for( int q = 0; q < H; ++q )
				{
				const double a = fold( sumk[q] ), b = fold( summ[q] );
				dst[lane + 64 * q] = a;             stage[lane + 64 * q] = a;
				dst[C - lane - 64 * q] = b;         stage[C - lane - 64 * q] = b;
				}
			const double vx = fold( sumx );
			if( lane == 0 ) { dst[C / 2] = vx; stage[C / 2] = vx; }
			}
		const bool any_bad = __any( bad );
		if( p.nan_out && lane == 0 && active )
			{
			if( chain == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }   // [4]: the sums of this epoch are in the workspace
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
		if( p.group_sums )
			{
			__syncthreads();
			const int live = min( WAVES, p.chains_per_channel - group * WAVES );      // wavefronts of this group that walked a chain
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
// PV::convert_to_audio (Conversions/AudioPV.cpp:86-139), dft 1024 / 512; HOPQ = hop / 128 (overlap-add accumulator in registers: the window a
// multiple of 128 samples).  k_synthesize_v2 at the smaller sizes.
// =================================================================================================================
template<int LOG2C, int WAVES, int HOPQ, int OCC>
__global__ __launch_bounds__( 64 * WAVES, OCC ) void k_synthesize_v3( SynthParams p )
	{
	using L = V3Lds<LOG2C>;
	constexpr int C = L::C, E = L::E, H = E / 2, NT = 64 * WAVES, QS = L::QS;
	static_assert( HOPQ == 0 || HOPQ == 1 || HOPQ == 2 || HOPQ == 4 || HOPQ == 8, "hop 128 / 256 / 512 / 1024, or 0: any hop <= window, any window <= dft" );
	static_assert( HOPQ <= E, "hop <= dft" );
	// RING (HOPQ = 0): the overlap-add accumulator as a ring of `window` floats per wavefront in LDS instead of registers -- hops that are no multiple of
	// 128, windows that are none: (1000, 250, 1024) ran the generic synthesis until round 5.  Samples leave one by one (4-byte stores), the chains'
	// overlaps through k_ola_fixup.
	constexpr bool RING = HOPQ == 0;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 );   // (a scalar: frame ranges, loop counts and base addresses live in SGPRs, the loops are scalar loops)
	const int W = p.window_size;
	const int cancel_word = cancel_peek( p.cancel );
	v3_load_twiddles<LOG2C, NT>( s + L::TW, p.tw, tid );
	for( int i = tid; i < C / 2; i += NT ) { const cf w = p.tw2[i]; s[L::W2 + i] = mk( w.x, -w.y ); }   // exp( +2 pi i k / 2C )
		{
		float * win = reinterpret_cast<float*>( s + L::WIN );
		for( int i = tid; i < 2 * C; i += NT ) win[i] = ( i < W ) ? p.window[i] * p.window_scale : 0.0f;      // AudioPV.cpp:102
		}
	if( !p.group_sums ) __syncthreads();                                        // (with group sums the barrier of the carry prologue below serves the tables too)
	const cf * s_tw = s + L::TW;
	const cf * s_w2 = s + L::W2 + lane;
	const cf * s_win = s + L::WIN + lane;
	cf * buf = s + L::buf0( false ) + wave * L::BUF_LEN;

	const int groups = ( p.chains_per_channel + WAVES - 1 ) / WAVES;
	const int channel = int( blockIdx.x ) / groups, group = int( blockIdx.x ) % groups;
	const int chain_in_channel_raw = group * WAVES + wave;
	const bool active = chain_in_channel_raw < p.chains_per_channel && !cancel_seen( cancel_word );
	const int chain_in_channel = active ? chain_in_channel_raw : 0;
	const int64_t chain = int64_t( channel ) * p.chains_per_channel + chain_in_channel;
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	const int hop = RING ? p.hop : 128 * HOPQ;
	const int wpad = ( W + 63 ) & ~63;
	float * ring = reinterpret_cast<float*>( s + L::buf0( false ) + WAVES * L::BUF_LEN ) + wave * wpad;      // RING only
	if constexpr( RING ) { for( int i = lane; i < wpad; i += 64 ) ring[i] = 0.0f; }
	int ring_base = 0;
	float * out1 = p.out + int64_t( channel ) * p.out_len;
	cf * out2 = reinterpret_cast<cf*>( out1 );
	cf * head2 = reinterpret_cast<cf*>( p.head + chain * p.head_len );
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;
	const int64_t tb0 = int64_t( group ) * WAVES * p.L;
	const int relf0 = ( active ? wave : 0 ) * p.L;                              // this chain's first frame, counted from tb0
	const int nf = int( t1 - t0 );
	const char * const pvb = reinterpret_cast<const char*>( p.pv + ( int64_t( channel ) * p.F + tb0 ) * ( C + 1 ) );
	char * const ob = reinterpret_cast<char*>( out1 ) + ( int64_t( hop ) * tb0 - W / 2 ) * 4;   // (in front of the buffer for a channel's first group: its chain 0 takes the general emit_step)
	const unsigned lane8 = 8u * unsigned( lane );
	const bool plain_emit = __builtin_amdgcn_readfirstlane( int( chain_in_channel != 0 ) ) != 0;
	const int padl = lane + lane / E;
	cf * mirror = buf + ( C - 64 ) / E * ( E + 1 ) + ( 64 - lane ) + ( 64 - lane ) / E;      // mirror[-QS q] = slot PAD( C - lane - 64 q )

	const DivC k_ard = p.ar_div;
	cf acc[E];                                                                  // overlap-add accumulator: acc[q] <-> samples pos + 128 q + 2 lane (+1)
	#pragma unroll
	for( int q = 0; q < E; ++q ) acc[q] = mk( 0.0f, 0.0f );

	// one 128-sample step leaves the chain; exactly one store instruction per step, never inside a branch (see k_synthesize_v2)
	cf * dump2 = reinterpret_cast<cf*>( p.dump ) + lane;
	int64_t pos_of_ring = chain_start;                                           // RING: ring[ring_base] <-> this absolute sample
	const bool fix = !RING && p.fix_state != nullptr;                           // this launch adds the chains' overlaps itself (below; see k_synthesize_v2)
	float * head1 = p.head + chain * p.head_len;
	// RING: the oldest `count` samples of the ring leave (and are cleared); sample by sample like the generic kernels (pv_kernels.h)
	auto ring_emit = [&]( int count, bool clear )
		{
		if( clear && count <= W && pos_of_ring >= own_start && pos_of_ring >= 0 && pos_of_ring + count <= p.out_len )
			{
			// the whole step lies in the output proper (every step past the chain's head but a channel's very first and last): a scalar base, no
			// per-sample routing
			float * const dst = out1 + pos_of_ring;
			for( int e = lane; e < count; e += 64 )
				{
				int j = ring_base + e; if( j >= W ) j -= W;
				dst[e] = ring[j]; ring[j] = 0.0f;
				}
			return;
			}
		for( int e = lane; e < count; e += 64 )
			{
			float v = 0.0f;
			if( e < W )
				{
				int j = ring_base + e; if( j >= W ) j -= W;
				v = ring[j]; if( clear ) ring[j] = 0.0f;
				}
			const int64_t a = pos_of_ring + e;
			if( a < own_start ) head1[a - chain_start] = v;
			else if( a >= 0 && a < p.out_len ) out1[a] = v;
			}
		};
	auto emit_step = [&]( int64_t a0, cf v )
		{
		const int64_t a = a0 + 2 * lane;
		cf * dst = ( a0 < own_start ) ? head2 + ( ( a - chain_start ) >> 1 ) : out2 + ( a >> 1 );
		if( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) dst = dump2;
		if( fix && a0 < own_start ) st_agent( dst, v );                          // (the head another wavefront may come to add up)
		else *dst = v;
		};
	cf mfk[H], mfm[H], mfx;
	auto load_row = [&]( int fr )                                               // fr: the frame, counted from tb0
		{
		const unsigned ro = unsigned( fr ) * unsigned( ( C + 1 ) * 8 );
		const cf * row = reinterpret_cast<const cf*>( pvb + ro );
		const cf * rowk = reinterpret_cast<const cf*>( pvb + ( ro + lane8 ) );
		const cf * rowm = reinterpret_cast<const cf*>( pvb + ( ro + unsigned( C * 8 ) - lane8 ) );
		#pragma unroll
		for( int q = 0; q < H; ++q )
			{
			mfk[q] = __builtin_nontemporal_load( rowk + 64 * q );
			mfm[q] = __builtin_nontemporal_load( rowm - 64 * q );
			}
		mfx = __builtin_nontemporal_load( row + C / 2 );
		};

	// phase_buffer (AudioPV.cpp:105) on entry to the chain, of the lane's pairs and of bin C/2
	double phk[H], phm[H], phx;
	if( p.group_sums )
		{
		// the carry prologue of k_synthesize_v2: group_carry (or the totals of the groups before this one) + the chains of this group before this chain
		auto fold = []( double r )
			{
			return ( __builtin_fabs( r ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( r ) : fold_phase_any( r );
			};
		const double * gs = ( p.group_carry ? p.group_carry : p.group_sums ) + int64_t( channel ) * groups * ( C + 1 );
		const double * sums0 = p.carry + ( int64_t( channel ) * p.chains_per_channel + int64_t( group ) * WAVES ) * ( C + 1 );
		const int live = min( WAVES, p.chains_per_channel - group * WAVES );
		constexpr int NB = ( C + NT ) / NT;                                       // bins per thread
		int bins_of[NB]; bool has[NB]; double run[NB];
		#pragma unroll
		for( int b = 0; b < NB; ++b ) { bins_of[b] = tid + NT * b; has[b] = bins_of[b] <= C; if( !has[b] ) bins_of[b] = C; run[b] = 0.0; }
		double vc[NB][WAVES];
		#pragma unroll
		for( int b = 0; b < NB; ++b )
			{
			#pragma unroll
			for( int w = 0; w < WAVES; ++w ) vc[b][w] = ( w < live ) ? sums0[int64_t( w ) * ( C + 1 ) + bins_of[b]] : 0.0;
			}
		if( p.group_carry )
			{
			#pragma unroll
			for( int b = 0; b < NB; ++b ) run[b] = gs[int64_t( group ) * ( C + 1 ) + bins_of[b]];
			}
		if( active ) load_row( relf0 );                                           // the first MF row travels while the carries are worked out
		if( !p.group_carry )
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
					for( int b = 0; b < NB; ++b ) run[b] = fold( run[b] + v[b][u] );
					}
				}
			}
		#pragma unroll
		for( int w = 0; w < WAVES; ++w )
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
			if( p.skip_words ) const_cast<int*>( p.skip_words )[4] = 0;              // a handed-over pre-pass is good for one convert_to_audio (k_sums_and_groups has read the word: a launch ago)
			}
		__syncthreads();
		const double * mine = reinterpret_cast<const double*>( buf );
		#pragma unroll
		for( int q = 0; q < H; ++q ) { phk[q] = mine[lane + 64 * q]; phm[q] = mine[C - lane - 64 * q]; }
		phx = mine[C / 2];
		wave_sync();
		}
	else if( active )
		{
		const double * carry = p.carry + chain * ( C + 1 );
		#pragma unroll
		for( int q = 0; q < H; ++q ) { phk[q] = carry[lane + 64 * q]; phm[q] = carry[C - lane - 64 * q]; }
		phx = carry[C / 2];
		}
	if( !active ) return;
	cf z[E];
	// inverse phase vocoder of the row in mfk / mfm / mfx (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61), merge of X[0..C] into the
	// half-size spectrum: leaves z[] complete
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
			if( q == 0 ) { a.y = ( lane == 0 ) ? 0.0f : a.y; b.y = ( lane == 0 ) ? 0.0f : b.y; }   // c2r ignores Im X[0], Im X[C]
			const cf w = s_w2[64 * q];
			const float ax = a.x + b.x, ay = a.y - b.y;                         // A = X[k] + conj X[C-k]
			const float dx2 = a.x - b.x, dy = a.y + b.y;                        // D = X[k] - conj X[C-k]
			const float bx = __builtin_fmaf( w.x, dx2, -( w.y * dy ) ), by = __builtin_fmaf( w.x, dy, w.y * dx2 );
			z[q] = mk( ax - by, -( ay + bx ) );                                 // Z[k]
			mirror[-QS * q] = mk( ax + by, ay - bx );                           // Z[C-k] (lane 0, q = 0: an unused slot)
			}
		if( lane == 0 ) buf[C / 2 + C / 2 / E] = mk( 2.0f * xx.x, 2.0f * xx.y );   // Z[C/2] = 2 X[C/2]
		wave_sync();
		#pragma unroll
		for( int q = H; q < E; ++q ) z[q] = buf[padl + QS * q];
		wave_sync();
		};

	if( !p.group_sums ) load_row( relf0 );
	bins_of_row();
	int64_t pos = chain_start;
	int rel = 0;
	// the overlaps of neighbouring chains added by the chains themselves (p.fix_state set): k_synthesize_v2's protocol, statement for statement --
	// a tagged word per boundary, the head's owner publishing from inside its frame loop, the tail's owner reading the word a frame before its last
	// and requesting the head under its last transform
	const int tag_tail = p.fix_tag | 1, tag_head = p.fix_tag | 2;
	const int nsteps = p.head_len / 128;
	const bool has_head = chain_in_channel != 0, has_tail = !last_chain;
	int * const word_h = p.fix_state + chain, * const word_t = p.fix_state + ( chain + 1 );      // (used under `fix` only)
	const cf * const head_next = reinterpret_cast<const cf*>( p.head + ( chain + 1 ) * p.head_len ) + lane;
	const int i_pub = ( p.head_len + hop - 1 ) / hop;
	int old_h = 0, seen_t = 0;
	bool published = false;
	cf hx[E];
	auto frame_step = [&]( int i, auto last_tag ) -> bool
		{
		constexpr bool LAST = decltype( last_tag )::value;
		bool have_head = false;
		if constexpr( LAST )
			{
			if( fix && has_tail )
				{
				have_head = __builtin_amdgcn_readfirstlane( seen_t ) == tag_head;
				if( have_head )
					{
					#pragma unroll
					for( int q = 0; q < E; ++q ) hx[q] = ( q < nsteps ) ? ld_agent( head_next + 64 * q ) : mk( 0.0f, 0.0f );
					}
				}
			}
		else
			{
			load_row( relf0 + i + 1 );
			if( fix && has_tail && i == nf - 2 && lane == 0 ) seen_t = __hip_atomic_load( word_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
			}
		fft_v3<LOG2C>( z, buf, s_tw, lane );
		// ---- G = fft( conj Z ): x[2n] = G[n].x, x[2n+1] = -G[n].y; window and overlap-add (AudioPV.cpp:122-134)
		if constexpr( RING )
			{
			#pragma unroll
			for( int q = 0; q < E; ++q )
				{
				const cf w = s_win[64 * q];
				const int s0 = 2 * ( lane + 64 * q );
				if( s0 < W )
					{
					// (a read, an addition and a write per sample: the LDS's own float add, ds_add_f32, was measured THREE times slower here)
					int j = ring_base + s0; if( j >= W ) j -= W;
					ring[j] += z[q].x * w.x;
					if( s0 + 1 < W ) { int j1 = j + 1; if( j1 >= W ) j1 -= W; ring[j1] += ( -z[q].y ) * w.y; }
					}
				}
			wave_sync();
			ring_emit( hop, true );
			wave_sync();
			pos_of_ring += hop;
			ring_base = ( hop < W ) ? ring_base + hop : 0;
			if( ring_base >= W ) ring_base -= W;
			pos += hop;
			rel += hop;
			if constexpr( !LAST ) bins_of_row();
			return false;
			}
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const cf w = s_win[64 * q];                                         // zero beyond W
			acc[q].x += z[q].x * w.x;
			acc[q].y += ( -z[q].y ) * w.y;
			}
		if( plain_emit && rel >= p.head_len )
			{
			const unsigned oo = unsigned( hop * ( relf0 + i ) ) * 4u + lane8;
			#pragma unroll
			for( int q = 0; q < HOPQ; ++q ) *reinterpret_cast<cf*>( ob + oo + 512 * q ) = acc[q];
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < HOPQ; ++q ) emit_step( pos + 128 * q, acc[q] );
			}
		#pragma unroll
		for( int q = 0; q < E; ++q ) acc[q] = ( q + HOPQ < E ) ? acc[q + HOPQ] : mk( 0.0f, 0.0f );
		pos += hop;
		rel += hop;
		if constexpr( !LAST )
			{
			bins_of_row();
			if( fix && has_head && i == i_pub )
				{
				publish_drain();                                                 // (the head's stores have retired: an explicit drain, once per chain -- pv_kernels_v2.h)
				if( lane == 0 ) old_h = __hip_atomic_exchange( word_h, tag_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
				asm volatile( "" ::: "memory" );
				published = true;
				}
			}
		return have_head;
		};
	for( int i = 0; i + 1 < nf; ++i ) frame_step( i, std::false_type{} );
	const bool have_head = frame_step( nf - 1, std::true_type{} );
	// flush the partial sums that the next chain's head completes; the last chain zero-fills to the end of the output
	const int64_t ring_end = pos + ( W - hop );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	if constexpr( RING )
		{
		for( int64_t a = pos + lane; a < flush_end; a += 64 )
			{
			float v = 0.0f;
			if( a < ring_end )
				{
				int j = ring_base + int( a - pos ); if( j >= W ) j -= W;
				v = ring[j];
				}
			if( a < own_start ) head1[a - chain_start] = v;
			else if( a >= 0 && a < p.out_len ) out1[a] = v;
			}
		return;
		}
	if( !fix || last_chain )
		{
		#pragma unroll
		for( int q = 0; q < E; ++q )
			{
			const int64_t a0 = pos + 128 * q;
			if( a0 < flush_end ) emit_step( a0, acc[q] );
			}
		for( int64_t a0 = pos + 128 * E; a0 < flush_end; a0 += 128 ) emit_step( a0, mk( 0.0f, 0.0f ) );
		}
	if( fix )
		{
		if( has_head && !published )
			{
			asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
			if( lane == 0 ) old_h = __hip_atomic_exchange( word_h, tag_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
			}
		if( has_tail )
			{
			cf * tail_next = reinterpret_cast<cf*>( p.tail + ( chain + 1 ) * p.head_len ) + lane;
			bool add = have_head;
			if( !add )
				{
				#pragma unroll
				for( int q = 0; q < E; ++q ) if( q < nsteps ) st_agent( tail_next + 64 * q, acc[q] );
				asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				int old = 0;
				if( lane == 0 ) old = __hip_atomic_exchange( word_t, tag_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
				add = __builtin_amdgcn_readfirstlane( old ) == tag_head;
				if( add )
					{
					#pragma unroll
					for( int q = 0; q < E; ++q ) hx[q] = ( q < nsteps ) ? ld_agent( head_next + 64 * q ) : mk( 0.0f, 0.0f );
					}
				}
			if( add )
				{
				#pragma unroll
				for( int q = 0; q < E; ++q )
					{
					const int64_t a = pos + 128 * q + 2 * lane;
					if( q < nsteps && a >= 0 && a < p.out_len ) out2[a >> 1] = mk( acc[q].x + hx[q].x, acc[q].y + hx[q].y );
					}
				}
			}
		if( has_head && __builtin_amdgcn_readfirstlane( old_h ) == tag_tail )
			{
			const cf * tl = reinterpret_cast<const cf*>( p.tail + chain * p.head_len ) + lane;
			const cf * hd = head2 + lane;
			#pragma unroll 4
			for( int q = 0; q < nsteps; ++q )
				{
				const cf t = ld_agent( tl + 64 * q ), h = ld_agent( hd + 64 * q );
				const int64_t a = chain_start + 128 * q + 2 * lane;
				if( a >= 0 && a < p.out_len ) out2[a >> 1] = mk( t.x + h.x, t.y + h.y );
				}
			}
		}
	}

} // namespace flanhip
