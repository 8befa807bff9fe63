// pv_kernels_eo.h -- dft 4096 (the reference API's default dft_size) with window <= 2048: analysis as TWO 1024-point register transforms
// per frame and wavefront instead of one 2048-point one.
//
// The round-1 kernel for this size (pv_kernels_fast.h, LOG2C = 11) keeps the 2048 complex points of a frame in registers: 32 per lane,
// ~400 registers with the per-bin state, one wavefront per SIMD, and 17.4 KB of LDS per wavefront beside 49 KB of tables -- nothing
// hides its LDS round trips and memory waits (0.23 of the HBM roofline against the dft 2048 kernels' 0.38).  Here the frame's packed
// complex sequence c[n] = x[2n] + i x[2n+1] (n < 2048, zero from W/2 on) is split by parity:
//        E = FFT1024( c[2m] ),  O = FFT1024( c[2m+1] ),      Z[k] = E[k] + w^k O[k],   Z[k+1024] = E[k] - w^k O[k],   w = exp( -2 pi i / 2048 )
// each half the dft 2048 kernel's own register transform (fft_fast<10>, 16 points per lane, the upper half of its input zero because
// the window is at most half the transform).  A lane owns the QUADS  k = lane + 64 q, q < 8:  from E[k], O[k] (its own registers) and
// E[1024-k], O[1024-k] (the mirror lane's, through LDS) come Z[k], Z[1024+k], Z[1024-k], Z[2048-k], i.e. the two mirror pairs
// ( k, 2048-k ) and ( 1024-k, 1024+k ) of the real-transform split: 4 bins per quad, with the dft 2048 kernel's per-bin
// code (polar_v, exact divisions, pv_kernels_v2.h).  The two transforms of a frame belong to the two wavefronts of a TEAM (below), each
// then owning half the quads: per frame a wavefront does what a dft 2048 wavefront does, with its register budget, and blocks of
// 8 wavefronts = 4 teams = 2 wavefronts per SIMD fit (160 KB of LDS exactly: two E / O buffer sets per team and 24 KB of tables).
//   lane 0, q = 0: the quad degenerates to bins 0, 2048 (from Z[0]) and 1024 (= conj Z[1024], computed twice);
//   k = 512: bins 512 and 1536, one pair, carried redundantly by every lane and stored by lane 0.
#pragma once
#include "pv_kernels_v2.h"

namespace flanhip {

struct EoLds
	{
	static constexpr int C = 1024;
	static constexpr int TW1 = 0;                          // [15][16]
	static constexpr int TW3 = TW1 + 240;                  // [3][256] for 1024 points
	static constexpr int TWQ = TW3 + 768;                  // [512] float4 { 0.5 cos, -0.5 sin of 2 pi k / 4096 ; cos, -sin of 2 pi k / 2048 }
	static constexpr int WIN = TWQ + 1024;                 // [2048 floats], zero beyond W
	static constexpr int BUF = WIN + 1024;
	static constexpr int BUF_LEN = C + C / 16 + 1;         // one half transform; a wavefront has two
	static constexpr size_t bytes( int waves ) { return size_t( BUF + waves * 2 * BUF_LEN ) * 8; }
	};
// the same with a window table of 4096 floats (windows above 2048: WBIG variants of the team kernels, one buffer set per team)
struct EoLdsBig
	{
	static constexpr int C = 1024;
	static constexpr int TW1 = 0;
	static constexpr int TW3 = TW1 + 240;
	static constexpr int TWQ = TW3 + 768;
	static constexpr int WIN = TWQ + 1024;                 // [4096 floats], zero beyond W
	static constexpr int BUF = WIN + 2048;
	static constexpr int BUF_LEN = C + C / 16 + 1;
	static constexpr size_t bytes( int waves ) { return size_t( BUF + waves * 2 * BUF_LEN ) * 8; }
	};

// the real-transform split of one mirror pair ( j, N - j ): X[j] and X[N-j] from Z[j] = zk, Z[N-j] = zm and w = 0.5 exp( -2 pi i j / 2N )
__device__ __forceinline__ void split_pair( cf zk, cf zm, float wx, float wy, float & rk, float & ik, float & rm, float & im )
	{
	const float sx = zk.x + zm.x, dy = zk.y + zm.y, dx = zk.x - zm.x, sy = zk.y - zm.y;
	const float t1v = __builtin_fmaf( wx, dy, wy * dx );
	const float t2v = __builtin_fmaf( wx, dx, -( wy * dy ) );
	rk = __builtin_fmaf( 0.5f, sx, t1v );  ik = __builtin_fmaf( 0.5f, sy, -t2v );
	rm = __builtin_fmaf( 0.5f, sx, -t1v ); im = __builtin_fmaf( -0.5f, sy, -t2v );
	}

// =================================================================================================================
// A TEAM of two wavefronts per chain: wavefront 0 of a team transforms the even points and then owns the
// quads k < 256, wavefront 1 the odd points and the quads 256 <= k < 512 (and k = 512).  Per frame a wavefront does exactly what the
// dft 2048 kernel's wavefront does -- one 1024-point register transform and 16 bins -- so its registers (16 previous phases, 16 chain
// sums) fit the fused round trip's sums without spilling.  (Round 2 also had a one-wavefront-per-chain form of the same decomposition for
// the plain convert_to_PV: 32 previous phases per lane, 256 VGPRs and 100-220 bytes of scratch reloaded inside the frame loop; once the
// team kernel had lost its splats and canonicalising maxima it was the faster one without sums too -- 0.280 against 0.307 ms at hop 512 --
// and the one-wavefront kernel was removed, round 3.)
// The two halves meet in the team's E / O buffers (TeamSync below; every team walks the same number of iterations, idle ones included).
__device__ __forceinline__ void lds_block_sync()
	{
	asm volatile( "s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory" );
	}
// The two wavefronts of ONE team meet; the other teams of the block are not involved (s_barrier would hold all eight wavefronts of the block
// to the pace of the slowest one, every frame).  A counter in the team's LDS (slot 1087 of its first buffer: the padded layout ends at 1086,
// the k = 0 lane's dummy mirror slot is 1088): each wavefront adds one when its LDS writes have landed, then polls until both have -- `target`
// = 2 x the number of meetings so far.  Both wavefronts of a team walk the same number of meetings (idle iterations included), both are
// resident (same block), so the wait ends; the poll count is bounded anyway, so that a logic error shows as wrong output, not as a hung GPU.
typedef __attribute__(( address_space( 3 ) )) unsigned lds_u32;
#ifndef FLANHIP_TEAM_SLEEP
#define FLANHIP_TEAM_SLEEP 1
#endif
// A team's stage of 2049 doubles (chain sums on their way to a group total; carries on their way to the team's two wavefronts), in cf slots
// relative to the team's first buffer.  Two buffer sets per team: analysis behind the sync word (slot 1087) and the dummy mirror slot, synthesis
// in the second set.  One set (WBIG): from slot 0, stepping over the sync word -- bins from 1087 on sit two slots higher.
template<bool ONE_SET> __device__ __forceinline__ int team_stage_slot( int bin, int base_two_sets )
	{
	if constexpr( ONE_SET ) return bin + ( bin >= 1087 ? 2 : 0 );
	else return base_two_sets + bin;
	}

// (an index the compiler cannot prove loop-invariant: the window loads of the ring form stay inside the frame loop instead of 32 registers)
__device__ __forceinline__ int eo_opaque( int v ) { asm volatile( "" : "+v"( v ) ); return v; }

struct TeamSync
	{
	lds_u32 * flag; unsigned target; int lane;
	__device__ __forceinline__ void meet()
		{
		target += 2;
		asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );
		if( lane == 0 ) (void) __hip_atomic_fetch_add( flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
		for( int spins = 0; spins < ( 1 << 22 ); ++spins )
			{
			const unsigned v = __builtin_amdgcn_readfirstlane( __hip_atomic_load( flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP ) );
			if( v >= target ) break;
			__builtin_amdgcn_s_sleep( FLANHIP_TEAM_SLEEP );
			}
		asm volatile( "" ::: "memory" );
		}
	};

// WBIG: windows up to 4096 (the whole transform): E and O inputs are no longer half zero -- 16 sample pairs per lane instead of 8, a 16 KB
// window table, which leaves LDS for ONE buffer set (two meetings per frame).  With the fused round trip's sums the kernel sits at 256
// VGPRs; what spills (48 bytes) spills inside the digital-silence arm of polar_v, not in the frame loop.
template<int TEAMS, bool SUMS, int QV = 2, bool DOUBLE = false, bool WBIG = false>   // DOUBLE: two E / O buffer sets per team, ONE barrier per frame
__global__ __launch_bounds__( 128 * TEAMS ) void k_analyze_eo_team( AnalyzeParams p, FastTables tb )
	{
	using L = typename std::conditional<WBIG, EoLdsBig, EoLds>::type;
	static_assert( !WBIG || !DOUBLE, "windows above 2048: one buffer set" );
	constexpr int C = 1024, N2 = 2048, Q = 4, NT = 128 * TEAMS;            // Q: quads per lane of ONE wavefront
	constexpr int WQ = WBIG ? 16 : 8, WMAX = 256 * WQ;                       // sample pairs per lane of one wavefront's half frame; the window this kernel admits
	typedef float VB __attribute__(( ext_vector_type( 4 * QV ) ));           // QV quads (4 bins each) as one vector stream
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	// the wavefront's number as a SCALAR (the compiler cannot see that threadIdx.x >> 6 is uniform): chain, frame range, role and the
	// audio pointer derived from it then live in scalar registers, the sample loads take a scalar base, the branches on them are scalar
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 ), team = wave >> 1, role = wave & 1;
	const int W = p.window_size, hop = p.hop;

	for( int i = tid; i < 240; i += NT ) s[L::TW1 + i] = tb.tw1[i];
	for( int i = tid; i < 768; i += NT ) s[L::TW3 + i] = tb.tw3[( 2 * ( i >> 8 ) + 1 ) * 256 + ( i & 255 )];
		{
		v4f_t * twq = reinterpret_cast<v4f_t*>( s + L::TWQ );
		for( int k = tid; k < 512; k += NT ) { const cf a = tb.w2[k], b = tb.w2[2 * k]; twq[k] = v4f_t{ 0.5f * a.x, 0.5f * a.y, b.x, b.y }; }
		float * win = reinterpret_cast<float*>( s + L::WIN );
		for( int i = tid; i < WMAX; i += NT ) win[i] = ( i < W ) ? p.window[i] : 0.0f;            // AudioPV.cpp:60,65
		}
	cf * const buf0 = s + L::BUF + team * ( DOUBLE ? 4 : 2 ) * L::BUF_LEN;     // set b: E at buf0 + 2 b BUF_LEN, O behind it
	TeamSync team_sync{ (lds_u32*) reinterpret_cast<unsigned*>( buf0 + 1087 ), 0u, lane };
	if( role == 0 && lane == 0 ) *team_sync.flag = 0u;
	__syncthreads();
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const v4f_t * s_twq = reinterpret_cast<const v4f_t*>( s + L::TWQ ) + lane + 256 * role;   // this wavefront's quads: k = lane + 64 ( 4 role + q )
	const v4f_t * s_win = reinterpret_cast<const v4f_t*>( s + L::WIN ) + lane;

	// a block is a GROUP: TEAMS consecutive chains of ONE channel (the last group of a channel may be short: its spare teams idle -- they still
	// walk their iterations and meet)
	const int gpc = ( p.chains_per_channel + TEAMS - 1 ) / TEAMS;
	const int gchannel = int( blockIdx.x ) / gpc, group = int( blockIdx.x ) % gpc;
	const int chain_in_channel_raw = group * TEAMS + team;
	const bool active = chain_in_channel_raw < p.chains_per_channel && !cancel_seen( cancel_peek( p.cancel ) );   // (cancelled: core.hip)
	const int64_t chain = active ? int64_t( gchannel ) * p.chains_per_channel + chain_in_channel_raw : 0;
	const int channel = int( chain / p.chains_per_channel );
	const int64_t t0 = int64_t( chain % p.chains_per_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37
	const int padl = lane + ( lane >> 4 );
	const int mir = ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );      // buf[mir - 68 q] = slot PAD( C - lane - 64 q )
	const int n32 = int( p.n );
	const float rdft = 1.0f / float( 2 * N2 );
	const float fk0 = float( lane + 256 * role );                             // k of this wavefront's first quad

	float prev[Q][4], prevs[2] = { 0.0f, 0.0f };
	#pragma unroll
	for( int q = 0; q < Q; ++q ) { prev[q][0] = 0.0f; prev[q][1] = 0.0f; prev[q][2] = 0.0f; prev[q][3] = 0.0f; }   // AudioPV.cpp:44
	// Fused round trip: the chain's sums of the phase increments convert_to_audio will integrate (phase_vocoder.cpp:57-58), one double per
	// bin as in k_analyze_v2 (a wavefront of a team owns 16 bins: 32 registers, which this kernel has -- the one-wavefront form did not, and
	// an earlier version of this one kept a telescoped float residue per bin instead, at nine fp64-class instructions per bin and frame)
	double sm[SUMS ? Q : 1][4], sm2[2] = { 0.0, 0.0 };
	if constexpr( SUMS )
		{
		#pragma unroll
		for( int q = 0; q < Q; ++q ) { sm[q][0] = 0.0; sm[q][1] = 0.0; sm[q][2] = 0.0; sm[q][3] = 0.0; }
		}
	unsigned mmax = 0u;                                                       // bit patterns, see k_analyze_v2
	// The pair ( 512, 1536 ) is the odd wavefront's alone and would cost each of its lanes ~140 instructions per frame for two bins -- on the
	// wavefront the other one waits for.  Instead lane ( i & 63 ) keeps E[512] and O[512] of the chain's i-th frame, and once per 64 frames
	// (and at the chain's end) the batch is worked off one frame per lane (k_analyze_v2 does the same with its bin C/2).
	cf ringE = mk( 0.0f, 0.0f ), ringO = mk( 0.0f, 0.0f );
	int fidx = 0;                                                             // the iteration bins_of_frame is working on

	const int64_t tfirst = t0 > 0 ? t0 - 1 : t0;
	const int frames = active ? int( t1 - tfirst ) : 0;                       // iterations with work (the halo frame included)
	const int iters = p.L + 1;                                                // what every team of every block walks
	// per frame (see k_analyze_v2): the plain loads take the whole transform's span of samples whatever the window (its table is zero beyond W), so the
	// span, not the window, has to lie inside the signal (round 5: any window, not WMAX alone -- (2000, 500, 4096) and (4000, 1000, 4096) ran every frame
	// through the clamped loads)
	auto frame_inside = [&]( int64_t t ) { return int64_t( hop ) * t - W / 2 >= 0 && int64_t( hop ) * t - W / 2 + WMAX <= p.n; };
	constexpr std::true_type inside{};
	constexpr std::false_type outside{};

	cf raw[WQ];                                                               // this wavefront's half of a frame: points lane + 64 q, q < WQ, of its parity
	auto run_chain = [&]()
		{
		struct __attribute__(( packed, aligned( 4 ) )) f2u { float x, y; };
		auto load_half = [&]( int64_t t, auto fast_tag )
			{
			constexpr bool FAST = decltype( fast_tag )::value;
			const int start = int( int64_t( hop ) * t - W / 2 ) + 2 * role;
			const float * xs = x + start + 4 * lane + 1024;                       // one address per lane: the eight loads reach -4096 .. +3072 bytes from it (immediates)
			#pragma unroll
			for( int q = 0; q < WQ; ++q )
				{
				const int a = start + 4 * ( lane + 64 * q );
				if constexpr( FAST )
					{
					const f2u v = *reinterpret_cast<const f2u*>( xs + ( 256 * q - 1024 ) );
					raw[q] = mk( v.x, v.y );
					}
				else
					{
					const int w0 = 4 * ( lane + 64 * q ) + 2 * role;                    // zero outside the signal (AudioPV.cpp:54-62) and beyond the window (:65)
					const float v0 = ( a >= 0 && a < n32 && w0 < W ) ? x[a] : 0.0f;
					const float v1 = ( a + 1 >= 0 && a + 1 < n32 && w0 + 1 < W ) ? x[a + 1] : 0.0f;
					raw[q] = mk( v0, v1 );
					}
				}
			};
		auto transform_frame = [&]( int set )
			{
			cf * mybuf = buf0 + ( 2 * set + role ) * L::BUF_LEN;
			cf z[16];
			#pragma unroll
			for( int q = 0; q < WQ; ++q )
				{
				const v4f_t wv = s_win[64 * q];
				z[q] = role ? mk( raw[q].x * wv.z, raw[q].y * wv.w ) : mk( raw[q].x * wv.x, raw[q].y * wv.y );
				}
			#pragma unroll
			for( int q = WQ; q < 16; ++q ) z[q] = mk( 0.0f, 0.0f );                  // (window at most half the transform: the first pass folds the zeros away)
			fft_fast<10>( z, mybuf, s_tw1, s_tw3, lane );
			#pragma unroll
			for( int q = 0; q < 4 * Q; ++q ) mybuf[padl + 68 * q] = z[q];              // natural order: slot PAD( lane + 64 q )
			};

		auto bins_of_frame = [&]( int64_t t, int64_t tn, int set, auto halo_tag, auto next_fast )
			{
			constexpr bool halo = decltype( halo_tag )::value;
			const cf * bufE = buf0 + 2 * set * L::BUF_LEN, * bufO = bufE + L::BUF_LEN;
			cf * row = reinterpret_cast<cf*>( p.out + ( int64_t( channel ) * p.F + t ) * ( N2 + 1 ) );
			const int k0 = lane + 256 * role;
			cf * row_a = row + k0;                                                // bin k          (+ 64 q)
			cf * row_b = row + ( N2 - k0 );                                       // bin 2048 - k   (- 64 q)
			cf * row_c = row + ( C - k0 );                                        // bin 1024 - k   (- 64 q)
			cf * row_d = row + ( C + k0 );                                        // bin 1024 + k   (+ 64 q)
			const cf * ownE = bufE + padl + 272 * role, * ownO = bufO + padl + 272 * role;          // slot PAD( lane + 64 ( 4 role + q ) ) = padl + 68 ( 4 role + q )
			const cf * mirE = bufE + mir - 272 * role, * mirO = bufO + mir - 272 * role;
			#pragma unroll
			for( int g = 0; g < Q / QV; ++g )
				{
				if( g == Q / QV / 2 ) load_half( tn, next_fast );                        // the next frame's half travels during the second half of the bins
				VB re, im, pv, binf;
				#pragma unroll
				for( int i = 0; i < QV; ++i )
					{
					const int q = QV * g + i;
					const cf ek = ownE[68 * q], ok = ownO[68 * q];
					const cf em = mirE[-68 * q], om = mirO[-68 * q];                  // k = 0 (role 0, lane 0, q = 0) reads an unused slot: overridden below
					const v4f_t tw = s_twq[64 * q];
					const float px = __builtin_fmaf( tw.z, ok.x, -( tw.w * ok.y ) ), py = __builtin_fmaf( tw.z, ok.y, tw.w * ok.x );
					const float qx = __builtin_fmaf( tw.z, om.x, tw.w * om.y ), qy = __builtin_fmaf( tw.z, om.y, -( tw.w * om.x ) );
					const cf zk = mk( ek.x + px, ek.y + py ), zk2 = mk( ek.x - px, ek.y - py );
					const cf zm = mk( em.x - qx, em.y - qy ), zm2 = mk( em.x + qx, em.y + qy );
					float r0, i0, r1, i1, r2, i2, r3, i3;
					split_pair( zk, zm2, tw.x, tw.y, r0, i0, r1, i1 );
					split_pair( zm, zk2, -tw.y, -tw.x, r2, i2, r3, i3 );
					if( q == 0 )
						{
						const bool l0 = k0 == 0;
						const cf z0 = mk( ek.x + ok.x, ek.y + ok.y ), zc = mk( ek.x - ok.x, ek.y - ok.y );   // Z[0], Z[1024] where k = 0
						r0 = l0 ? z0.x + z0.y : r0;  i0 = l0 ? 0.0f : i0;               // X[0]
						r1 = l0 ? z0.x - z0.y : r1;  i1 = l0 ? 0.0f : i1;               // X[2048]
						r2 = l0 ? zc.x : r2;  i2 = l0 ? -zc.y : i2;                     // X[1024] = conj Z[1024], twice
						r3 = l0 ? zc.x : r3;  i3 = l0 ? -zc.y : i3;
						}
					re[4 * i + 0] = r0; im[4 * i + 0] = i0; re[4 * i + 1] = r1; im[4 * i + 1] = i1;
					re[4 * i + 2] = r2; im[4 * i + 2] = i2; re[4 * i + 3] = r3; im[4 * i + 3] = i3;
					const float fk = fk0 + float( 64 * q );
					binf[4 * i + 0] = fk * p.sample_rate * rdft;                                    // PVBuffer.cpp:443-446
					binf[4 * i + 1] = ( float( N2 ) - fk ) * p.sample_rate * rdft;
					binf[4 * i + 2] = ( float( C ) - fk ) * p.sample_rate * rdft;
					binf[4 * i + 3] = ( float( C ) + fk ) * p.sample_rate * rdft;
					#pragma unroll
					for( int j = 0; j < 4; ++j ) pv[4 * i + j] = prev[q][j];
					}
				VB phase, m;
				polar_v( re, im, phase, m );                                        // phase_vocoder.cpp:37-52 (AudioPV.cpp:69-73)
				#pragma unroll
				for( int i = 0; i < QV; ++i )
					{
					#pragma unroll
					for( int j = 0; j < 4; ++j ) prev[QV * g + i][j] = phase[4 * i + j];            // :45
					}
				if constexpr( !halo )
					{
					const VB expd = div_c_each( binf, p.ar_div ) * vsplat<VB>( FLANHIP_PI2_F );       // :47
					const VB delta_phase = ( phase - pv ) - expd;                                // :44, :47-48
					VB wrapped = delta_phase;
					if( use_wrapping ) wrapped = delta_phase - vsplat<VB>( FLANHIP_PI2_F ) * round_half_away_v( div_pi2_v( delta_phase ) );   // :39-42,49
					VB war;                                                                  // (element by element: no SGPR copy per element of a splat)
					#pragma unroll
					for( int i = 0; i < int( sizeof( VB ) / sizeof( float ) ); ++i ) war[i] = wrapped[i] * p.analysis_rate;
					const VB f = binf + div_pi2_v( war );                                    // :50-52
					#pragma unroll
					for( int i = 0; i < QV; ++i )
						{
						const int q = QV * g + i;
						__builtin_nontemporal_store( cf{ m[4 * i + 0], f[4 * i + 0] }, row_a + 64 * q );
						__builtin_nontemporal_store( cf{ m[4 * i + 1], f[4 * i + 1] }, row_b - 64 * q );
						__builtin_nontemporal_store( cf{ m[4 * i + 2], f[4 * i + 2] }, row_c - 64 * q );
						__builtin_nontemporal_store( cf{ m[4 * i + 3], f[4 * i + 3] }, row_d + 64 * q );
						}
					if constexpr( SUMS )
						{
						const VB term = div_c_each( f, p.ar_div ) * vsplat<VB>( FLANHIP_PI2_F );      // phase_vocoder.cpp:57-58
						#pragma unroll
						for( int i = 0; i < QV; ++i )
							{
							#pragma unroll
							for( int j = 0; j < 4; ++j ) sm[QV * g + i][j] += double( term[4 * i + j] );
							}
						#pragma unroll
						for( int i = 0; i < 4 * QV; i += 2 ) mmax = max( mmax, max( __float_as_uint( m[i] ), __float_as_uint( m[i + 1] ) ) );
						}
					}
				}
			if( role == 1 )
				{
				const cf e512 = bufE[544], o512 = bufO[544];                          // slot PAD( 512 )
				const bool mine = lane == ( fidx & 63 );
				ringE = mine ? e512 : ringE;  ringO = mine ? o512 : ringO;
				}
			};

		// iteration i of a chain: frame tfirst + i (the first is the halo when t0 > 0).  Rotated like the dft 2048 kernel: the per-bin work of
		// frame i, then the transform of frame i + 1; barriers after each (all teams, every iteration, work or not)
		// the batch of the pair ( 512, 1536 ): lane j holds frame tb + j, j < nb (frame tfirst, the halo, only lends its phases)
		auto flush_orphans = [&]( int64_t tb, int nb )
			{
			const int64_t t = tb + lane;
			const bool valid = lane < nb && t >= t0;
			// Z[512] = E[512] - i O[512], Z[1536] = E[512] + i O[512]; bins 512 and 1536 are one pair
			const cf zk = mk( ringE.x + ringO.y, ringE.y - ringO.x ), zm = mk( ringE.x - ringO.y, ringE.y + ringO.x );
			float r[2], im2[2];
			split_pair( zk, zm, 0.35355339059327379f, -0.35355339059327379f, r[0], im2[0], r[1], im2[1] );
			cf * rowp = reinterpret_cast<cf*>( p.out + ( int64_t( channel ) * p.F + ( valid ? t : t0 ) ) * ( N2 + 1 ) );
			const int j0 = __builtin_amdgcn_readfirstlane( ( tb < t0 ) ? 1 : 0 ), j1 = __builtin_amdgcn_readfirstlane( nb );
			#pragma unroll
			for( int j = 0; j < 2; ++j )
				{
				const float phase = atan2_fast( im2[j], r[j] );
				float pvx = __shfl_up( phase, 1 );                                 // the frame before: the lane below, or the batch before
				pvx = ( lane == 0 ) ? prevs[j] : pvx;
				prevs[j] = __shfl( phase, nb - 1 );
				const float bx = float( j == 0 ? 512 : 1536 ) * p.sample_rate * rdft;
				const float delta_phase = ( phase - pvx ) - div_c( bx, p.ar_div ) * FLANHIP_PI2_F;
				const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * round_half_away( div_pi2( delta_phase ) ) : delta_phase;
				const float f = bx + div_pi2( wrapped * p.analysis_rate );
				const float m = magnitude_scaled( r[j], im2[j] );
				if( valid ) __builtin_nontemporal_store( mk( m, f ), rowp + ( j == 0 ? 512 : 1536 ) );
				if constexpr( SUMS )
					{
					const float term = div_c( f, p.ar_div ) * FLANHIP_PI2_F;         // phase_vocoder.cpp:57-58
					for( int l = j0; l < j1; ++l ) sm2[j] += double( __uint_as_float( __builtin_amdgcn_readlane( __float_as_uint( term ), l ) ) );   // frame order
					mmax = valid ? max( mmax, __float_as_uint( m ) ) : mmax;
					}
				}
			};

		if( frames > 0 ) { if( frame_inside( tfirst ) ) load_half( tfirst, inside ); else load_half( tfirst, outside ); transform_frame( 0 ); }
		team_sync.meet();
		auto iteration = [&]( int i, auto halo_tag )
			{
			const int set = DOUBLE ? ( i & 1 ) : 0;
			if( i < frames )
				{
				const int64_t t = tfirst + i, tn = min( t + 1, t1 - 1 );           // (the last frame requests itself again: nobody waits for it)
				fidx = i;
				if( frame_inside( tn ) ) bins_of_frame( t, tn, set, halo_tag, inside );
				else bins_of_frame( t, tn, set, halo_tag, outside );
				if( role == 1 && ( ( i & 63 ) == 63 || i == frames - 1 ) ) flush_orphans( t - ( i & 63 ), ( i & 63 ) + 1 );
				}
			// one buffer set: nobody may write the next frame's E / O before both halves have read this one's.  Two sets: the next frame goes
			// to the other set, whose readers (the frame before this one) passed the previous barrier
			if constexpr( !DOUBLE ) team_sync.meet();
			if( i + 1 < frames ) transform_frame( DOUBLE ? ( set ^ 1 ) : 0 );
			team_sync.meet();                                                     // the next frame's E / O are written
			};
		// The halo frame (of which only the phases are wanted) is iteration 0 of every chain but a channel's first: peeled, so that the loop
		// proper holds one kind of iteration (with both kinds inside, the chain sums met at a join after every frame and the compiler paid
		// a 64-bit register copy per sum and frame)
		if( t0 > 0 ) iteration( 0, std::true_type{} ); else iteration( 0, std::false_type{} );
		for( int i = 1; i < iters; ++i ) iteration( i, std::false_type{} );
		};
	Stamps st;                                                                  // (diagnostic builds: the wavefront's life, tools/wave_spans.py)
	st.init();
	run_chain();
	st.flush( lane );

	if constexpr( SUMS )
		{
		// the chain's sums, folded like phase_vocoder.cpp:59: what k_phase_sums2 would leave in the workspace
		bool bad = mmax >= 0x7f800000u;
		auto fold = [&]( double sq ) -> double
			{
			bad |= !( __builtin_fabs( sq ) <= 1.7976931348623157e308 );              // a NaN / Inf frequency poisons its sum
			return ( __builtin_fabs( sq ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( sq ) : fold_phase_any( sq );
			};
		// (with group totals wanted, see below: the sums also go to a stage in the team's buffers -- team_stage_slot -- idle now that both
		// wavefronts have passed the last meeting after their last read)
		double * const stage = p.group_sums ? reinterpret_cast<double*>( buf0 ) : nullptr;
		auto slot = [&]( int bin ) { return team_stage_slot<!DOUBLE>( bin, 1090 ); };
		if( active )
			{
			double * dst = p.sums + chain * ( N2 + 1 );
			#pragma unroll
			for( int q = 0; q < Q; ++q )
				{
				const int k = lane + 256 * role + 64 * q;
				const double a = fold( sm[q][0] ), b = fold( sm[q][1] ), c = fold( sm[q][2] ), d = fold( sm[q][3] );
				dst[k] = a; dst[N2 - k] = b; dst[C - k] = c;
				if( k != 0 ) dst[C + k] = d;                                         // (k = 0: the quad's fourth bin is bin 1024 again)
				if( stage )
					{
					stage[slot( k )] = a; stage[slot( N2 - k )] = b; stage[slot( C - k )] = c;
					if( k != 0 ) stage[slot( C + k )] = d;
					}
				}
			if( role == 1 && lane == 0 )
				{
				const double a = fold( sm2[0] ), b = fold( sm2[1] );
				dst[512] = a; dst[1536] = b;
				if( stage ) { stage[slot( 512 )] = a; stage[slot( 1536 )] = b; }
				}
			}
		const bool any_bad = __any( bad ) && active;
		if( p.nan_out && lane == 0 && active )
			{
			if( chain == 0 && role == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
			{
			if( p.group_sums )
				{
				// one total per group lets the synthesis kernel work out its own carries (k_analyze_v2 does the same with its groups of 8; no scan
				// kernel between the two)
				__syncthreads();
				const int live = min( TEAMS, p.chains_per_channel - group * TEAMS );      // teams of this group that walked a chain
				double * gdst = p.group_sums + ( int64_t( gchannel ) * gpc + group ) * ( N2 + 1 );
				for( int bin = tid; bin <= N2; bin += NT )
					{
					double run = 0.0;
					for( int w = 0; w < live; ++w )
						{
						const double v = run + reinterpret_cast<const double*>( s + L::BUF + w * ( DOUBLE ? 4 : 2 ) * L::BUF_LEN )[slot( bin )];
						run = ( __builtin_fabs( v ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( v ) : fold_phase_any( v );
						}
					gdst[bin] = run;
					}
				}
			}
		}
	}


// =================================================================================================================
// PV::convert_to_audio (Conversions/AudioPV.cpp:86-139), dft 4096, window <= 2048, hop 256 / 512 / 1024: the mirror image of
// k_analyze_eo_team.  A team of two wavefronts per chain.  Per frame each wavefront runs the inverse phase vocoder on its 16 bins (the
// quads k = lane + 64 ( 4 role + q ), q < 4: bins k, 2048-k, 1024-k, 1024+k), merges them into the half-size spectrum Zc (conjugated: the
// forward transform of it is the conjugate of the inverse one) at k, 1024+k, 1024-k, 2048-k and leaves
//        A[k] = Zc[k] + Zc[k+1024],   B[k] = ( Zc[k] - Zc[k+1024] ) w^k      ( and the same at 1024-k;  w = exp( -2 pi i / 2048 ) )
// in the team's buffers; after one LDS-only barrier wavefront 0 transforms A (1024 points: the EVEN output points G[2m], i.e. samples
// 4m, 4m+1) and wavefront 1 transforms B (the odd ones: samples 4m+2, 4m+3), of which only m < 512 lie inside the window.  Each keeps its
// half of the overlap-add accumulator in registers (16 floats per lane: sample 256 q + 4 lane + 2 role (+1)) and emits its half of the
// finished hop.  Two buffer sets: one barrier per frame.
// Hop 128 (HS = 0; the reference API's default hop): a finished hop is HALF a 256-sample step -- the lower 32 lanes of acc[0] -- and the
// accumulator moves on by 32 lanes: new acc[q] = { upper half of acc[q], lower half of acc[q+1] }, one v_permlane32_swap and one select
// per register.
// WBIG: windows up to 4096 -- all 1024 output points of each transform lie inside the window (16 accumulator pairs per lane instead of 8), the
// window table is 16 KB, which leaves LDS for ONE A / B buffer set (two meetings per frame).
// HS = -1 (round 5): ANY hop <= window and any window -- the overlap-add accumulator is a ring of W floats per team in LDS instead of registers (one A / B
// buffer set: the ring takes the second set's place; windows above 2048: three teams per block).  Both wavefronts add their windowed halves into the
// ring, meet, and between that meeting and the next (the one that guards the A / B buffers anyway) send off and clear the hop samples that are final.
// One wavefront's side of the chains' overlaps added inside a team synthesis kernel instead of by k_ola_fixup4 (round 6; k_synthesize_v2's protocol, whose
// words explain it: a tagged word per boundary, written by atomic exchange; the head's owner says so from inside its frame loop once the head's last store
// has been issued, behind a drained queue; the tail's owner looks a frame before its end and adds the head to its accumulator as it leaves, or deposits the
// tail; whoever finds the other side's tag adds -- tail + head, one addition per sample, k_ola_fixup4's bits).  Every WAVEFRONT of a team runs it for its own
// part of a boundary -- the samples STEP q + jl (+1) behind it, q < WQ, the same set in the tail's owner and in the head's -- under a word of its own
// (word wpc chain + role), so the team never meets over it.
template<int WQ, int STEP>
struct ChainOverlap
	{
	bool on, has_head, has_tail, published;
	int tag_tail, tag_head;
	int * word_h, * word_t;
	int old_h, seen_t;                                                          // (lane 0's: what the head word held before this chain's tag; what the tail word holds)
	__device__ __forceinline__ void init( const SynthParams & p, bool active, int64_t chain, int wpc, int role, bool first_chain, bool last_chain )
		{
		on = active && p.fix_state != nullptr;
		has_head = !first_chain; has_tail = !last_chain; published = false;
		tag_tail = p.fix_tag | 1; tag_head = p.fix_tag | 2;
		word_h = p.fix_state + ( wpc * chain + role ); word_t = p.fix_state + ( wpc * ( chain + 1 ) + role );      // (used under `on` only)
		old_h = 0; seen_t = 0;
		}
	// behind frame i of `frames`, its stores issued: head_done = every store of the chain's head has been issued
	__device__ __forceinline__ void after_frame( int i, int frames, bool head_done, int lane )
		{
		if( !on ) return;
		if( has_tail && i == frames - 2 && lane == 0 ) seen_t = __hip_atomic_load( word_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
		if( has_head && !published && head_done )
			{
			publish_drain();
			if( lane == 0 ) old_h = __hip_atomic_exchange( word_h, tag_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
			asm volatile( "" ::: "memory" );
			published = true;
			}
		}
	// the end of the chain (a channel's last chain: behind its flush).  acc: the partial sums of the W - hop samples from `pos` on
	__device__ __forceinline__ void finish( const SynthParams & p, const cf ( & acc )[WQ], int jl, int64_t chain, int64_t chain_start, int64_t pos, cf * out2, int lane )
		{
		if( !on ) return;
		if( has_head && !published )
			{
			// a chain too short to have published from its loop (a channel's last): now, behind a drained queue
			asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
			if( lane == 0 ) old_h = __hip_atomic_exchange( word_h, tag_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
			}
		if( has_tail )
			{
			// this chain's tail meets the next chain's head
			const cf * head_next = reinterpret_cast<const cf*>( p.head + ( chain + 1 ) * p.head_len );
			cf * tail_next = reinterpret_cast<cf*>( p.tail + ( chain + 1 ) * p.head_len );
			bool add = __builtin_amdgcn_readfirstlane( seen_t ) == tag_head;         // the neighbour's head was complete a frame ago
			if( !add )
				{
				// not yet: leave the tail where the neighbour will find it, BEHIND a drained queue, and say so; if its tag has appeared meanwhile the addition is ours after all
				#pragma unroll
				for( int q = 0; q < WQ; ++q ) { const int j = STEP * q + jl; if( j < p.head_len ) st_agent( tail_next + ( j >> 1 ), acc[q] ); }
				asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
				int old = 0;
				if( lane == 0 ) old = __hip_atomic_exchange( word_t, tag_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
				add = __builtin_amdgcn_readfirstlane( old ) == tag_head;
				}
			if( add )
				{
				#pragma unroll
				for( int q = 0; q < WQ; ++q )
					{
					const int j = STEP * q + jl;
					const int64_t a = pos + j;
					if( j < p.head_len && a >= 0 && a < p.out_len )
						{
						const cf h = ld_agent( head_next + ( j >> 1 ) );
						out2[a >> 1] = mk( acc[q].x + h.x, acc[q].y + h.y );
						}
					}
				}
			}
		if( has_head && __builtin_amdgcn_readfirstlane( old_h ) == tag_tail )
			{
			// this chain's head meets the previous chain's tail, which was there when the head's tag went out
			const cf * tl = reinterpret_cast<const cf*>( p.tail + chain * p.head_len );
			const cf * hd = reinterpret_cast<const cf*>( p.head + chain * p.head_len );
			for( int j = jl; j < p.head_len; j += STEP )
				{
				const cf t = ld_agent( tl + ( j >> 1 ) ), h = ld_agent( hd + ( j >> 1 ) );
				const int64_t a = chain_start + j;
				if( a >= 0 && a < p.out_len ) out2[a >> 1] = mk( t.x + h.x, t.y + h.y );
				}
			}
		}
	};

template<int TEAMS, int HS, bool WBIG = false>                               // HS = hop / 256; 0: hop 128; -1: any hop, ring accumulator
__global__ __launch_bounds__( 128 * TEAMS ) void k_synthesize_eo_team( SynthParams p, FastTables tb )
	{
	using L = typename std::conditional<WBIG, EoLdsBig, EoLds>::type;
	constexpr bool RING = HS < 0;
	// WINGLOB (the ring form at windows above 2048, round 5): the scaled window is read from memory, sample pair by sample pair, and the fourth team's
	// ring takes the window table's 16 KB of LDS -- four teams per block instead of three (three rings of 16 KB beside the buffers, not four: 151 KB), which
	// is what the ANALYSIS in front pays for too, its chains being cut to the synthesis' layout ((4000, 1000, 4096): 50 -> 60 M frames/s)
	constexpr bool WINGLOB = RING && WBIG;
	static_assert( !WINGLOB || TEAMS == 4, "the fourth ring sits in the window table's place" );
	constexpr int C = 1024, N2 = 2048, Q = 4, NT = 128 * TEAMS;
	const int hop = RING ? p.hop : HS ? 256 * HS : 128;
	constexpr bool DOUBLE = !WBIG && !RING;
	constexpr int WQ = WBIG ? 16 : 8, WMAX = 256 * WQ;                          // accumulator pairs per lane; the window this kernel admits
	static_assert( HS == -1 || HS == 0 || HS == 1 || HS == 2 || HS == 4, "hop 128 / 256 / 512 / 1024, or the ring" );
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 ), team = wave >> 1, role = wave & 1;
	const int W = p.window_size;

	for( int i = tid; i < 240; i += NT ) s[L::TW1 + i] = tb.tw1[i];
	for( int i = tid; i < 768; i += NT ) s[L::TW3 + i] = tb.tw3[( 2 * ( i >> 8 ) + 1 ) * 256 + ( i & 255 )];
		{
		// { cos, sin of 2 pi k / 4096 ; cos, -sin of 2 pi k / 2048 }
		v4f_t * twq = reinterpret_cast<v4f_t*>( s + L::TWQ );
		for( int k = tid; k < 512; k += NT ) { const cf a = tb.w2[k], b = tb.w2[2 * k]; twq[k] = v4f_t{ a.x, -a.y, b.x, b.y }; }
		float * win = reinterpret_cast<float*>( s + L::WIN );
		if constexpr( !WINGLOB ) { for( int i = tid; i < WMAX; i += NT ) win[i] = ( i < W ) ? p.window[i] * p.window_scale : 0.0f; }   // AudioPV.cpp:102
		}
	cf * const buf0 = s + L::BUF + team * ( DOUBLE ? 4 : 2 ) * L::BUF_LEN;      // set b: A at buf0 + 2 b BUF_LEN, B behind it
	// RING: the teams' overlap-add rings behind the buffers, Wr floats each (W rounded up to a whole number of wavefront rows)
	const int Wr = ( W + 63 ) & ~63;
	float * const ring = ( WINGLOB && team == 3 ) ? reinterpret_cast<float*>( s + L::WIN ) : reinterpret_cast<float*>( s + L::BUF + TEAMS * 2 * L::BUF_LEN ) + team * Wr;
	if constexpr( RING ) { for( int i = lane + 64 * role; i < Wr; i += 128 ) ring[i] = 0.0f; }
	TeamSync team_sync{ (lds_u32*) reinterpret_cast<unsigned*>( buf0 + 1087 ), 0u, lane };
	if( role == 0 && lane == 0 ) *team_sync.flag = 0u;
	__syncthreads();
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const v4f_t * s_twq = reinterpret_cast<const v4f_t*>( s + L::TWQ ) + lane + 256 * role;
	const v4f_t * s_win = reinterpret_cast<const v4f_t*>( s + L::WIN ) + lane;

	// a block is a GROUP of TEAMS consecutive chains of one channel (see k_analyze_eo_team); spare teams of a channel's last group idle
	const int gpc = ( p.chains_per_channel + TEAMS - 1 ) / TEAMS;
	const int gchannel = int( blockIdx.x ) / gpc, group = int( blockIdx.x ) % gpc;
	const int chain_in_channel_raw = group * TEAMS + team;
	const bool active = chain_in_channel_raw < p.chains_per_channel && !cancel_seen( cancel_peek( p.cancel ) );   // (cancelled: core.hip; every team still walks its iterations and meets)
	const int64_t chain = active ? int64_t( gchannel ) * p.chains_per_channel + chain_in_channel_raw : 0;
	const int channel = int( chain / p.chains_per_channel );
	const int chain_in_channel = int( chain % p.chains_per_channel );
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	const int frames = active ? int( t1 - t0 ) : 0;
	float * out1 = p.out + int64_t( channel ) * p.out_len;
	cf * out2 = reinterpret_cast<cf*>( out1 );
	cf * head2 = reinterpret_cast<cf*>( p.head + chain * p.head_len );
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;
	const int padl = lane + ( lane >> 4 );
	const int mir = ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );
	const int k0 = lane + 256 * role;                                           // k of this wavefront's first quad
	// the chains' overlaps added here (ChainOverlap above): the register forms only, where the workspace holds the words (p.fix_state)
	ChainOverlap<WBIG ? 16 : 8, 256> ov;
	ov.init( p, !RING && active, chain, 2, role, chain_in_channel == 0, last_chain );
	const bool fix = ov.on;

	// running phases (AudioPV.cpp:105) on entry to the chain: [q][0..3] = bins k, 2048-k, 1024-k, 1024+k; the odd wavefront also 512, 1536
	double ph[Q][4], phs[2] = { 0.0, 0.0 };
	if( !p.group_sums )
		{
		const double * carry = p.carry + chain * ( N2 + 1 );
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			const int k = k0 + 64 * q;
			ph[q][0] = carry[k]; ph[q][1] = carry[N2 - k]; ph[q][2] = carry[C - k]; ph[q][3] = carry[C + k];
			}
		if( role == 1 ) { phs[0] = carry[512]; phs[1] = carry[1536]; }
		}
	cf acc[WQ];                                                                 // acc[q] <-> samples pos + 256 q + 4 lane + 2 role (+1)
	#pragma unroll
	for( int q = 0; q < WQ; ++q ) acc[q] = mk( 0.0f, 0.0f );

	// one 256-sample step of finished (or partial) output: this wavefront's half of it.  One store per step, never inside a branch (lanes
	// outside the output store into the dump area): a static number of stores behind the row request, a counted wait (see k_synthesize_v2)
	cf * dump2 = reinterpret_cast<cf*>( p.dump ) + lane;
	const unsigned lane_off = 16u * unsigned( lane ) + 8u * unsigned( role );     // bytes of samples 4 lane + 2 role
	auto emit_step = [&]( int64_t a0, cf v )
		{
		const int64_t a = a0 + 4 * lane + 2 * role;
		cf * dst = ( a0 < own_start ) ? head2 + ( ( a - chain_start ) >> 1 ) : out2 + ( a >> 1 );
		if( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) dst = dump2;
		if( fix && a0 < own_start ) st_agent( dst, v );                          // (the head another wavefront may come to add up: the end of the kernel)
		else *dst = v;
		};
	// hop 128: one 128-sample half step -- the lanes 32 part .. 32 part + 31 of v hold samples a0 + 4 ( lane & 31 ) + 2 role (+1); the other lanes dump
	auto emit_half = [&]( int64_t a0, cf v, int part )
		{
		const int64_t a = a0 + 4 * ( lane & 31 ) + 2 * role;
		cf * dst = ( a0 < own_start ) ? head2 + ( ( a - chain_start ) >> 1 ) : out2 + ( a >> 1 );
		const bool idle = ( lane >> 5 ) != part;
		if( idle || ( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) ) dst = dump2;
		// (the head at agent scope -- but never the idle half's stores into the dump area: 32 lanes of every wavefront writing ONE line through to memory
		// made the hop 128 launch twice as long)
		if( fix && a0 < own_start ) { if( !idle ) st_agent( dst, v ); }
		else *dst = v;
		};
	auto rotate_half = []( float a, float b, bool low ) -> float          // lanes 0..31 <- a's lanes 32..63, lanes 32..63 <- b's lanes 0..31
		{
		const auto r = __builtin_amdgcn_permlane32_swap( __float_as_uint( a ), __float_as_uint( b ), false, false );   // r[0] = { a.lo, b.lo }, r[1] = { a.hi, b.hi }
		return __uint_as_float( low ? r[1] : r[0] );
		};
	cf mf[Q][4], mfs[2];
	auto load_row = [&]( int64_t t )
		{
		const cf * row = reinterpret_cast<const cf*>( p.pv + ( int64_t( channel ) * p.F + t ) * ( N2 + 1 ) );
		const cf * ra = row + k0, * rb = row + ( N2 - k0 ), * rc = row + ( C - k0 ), * rd = row + ( C + k0 );
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			mf[q][0] = __builtin_nontemporal_load( ra + 64 * q );
			mf[q][1] = __builtin_nontemporal_load( rb - 64 * q );
			mf[q][2] = __builtin_nontemporal_load( rc - 64 * q );
			mf[q][3] = __builtin_nontemporal_load( rd + 64 * q );
			}
		mfs[0] = __builtin_nontemporal_load( row + ( 512 + 1024 * 0 ) );          // (both wavefronts load them: a static number of loads; the even one ignores them)
		mfs[1] = __builtin_nontemporal_load( row + 1536 );
		};
	// Zc[j], Zc[N-j] from X[j] = a, X[N-j] = b and w = exp( +2 pi i j / 2N ) (k_synthesize_v2's merge)
	auto merge_pair = []( cf a, cf b, float wx, float wy, cf & zj, cf & zn )
		{
		const float ax = a.x + b.x, ay = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
		const float bx = __builtin_fmaf( wx, dx, -( wy * dy ) ), by = __builtin_fmaf( wx, dy, wy * dx );
		zj = mk( ax - by, -( ay + bx ) );
		zn = mk( ax + by, ay - bx );
		};
	// inverse phase vocoder of the row in mf[] (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61), merge, A / B into buffer set `set`
	auto bins_of_row = [&]( int set )
		{
		cf * bufA = buf0 + 2 * set * L::BUF_LEN, * bufB = bufA + L::BUF_LEN;
		bool slow = false;
		// f / analysis_rate of the wavefront's 16 bins under ONE test of the divisor's plan (see k_synthesize_v2)
		float dv[Q][4];
		if( p.ar_div.exact )
			{
			const float dc = p.ar_div.c, drc = p.ar_div.rc;
			#pragma unroll
			for( int q = 0; q < Q; ++q )
				{
				#pragma unroll
				for( int j = 0; j < 4; ++j ) { const float x = mf[q][j].y, q0 = x * drc; dv[q][j] = __builtin_fmaf( __builtin_fmaf( -q0, dc, x ), drc, q0 ); }   // pv_math.h: div_c
				}
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < Q; ++q )
				{
				#pragma unroll
				for( int j = 0; j < 4; ++j ) dv[q][j] = mf[q][j].y / p.ar_div.c;
				}
			}
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			#pragma unroll
			for( int j = 0; j < 4; ++j )
				{
				ph[q][j] += double( dv[q][j] * FLANHIP_PI2_F );                        // phase_vocoder.cpp:57-58
				slow |= !( __builtin_fabs( ph[q][j] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
				}
			}
		if( role == 1 )
			{
			#pragma unroll
			for( int j = 0; j < 2; ++j )
				{
				phs[j] += double( div_c( mfs[j].y, p.ar_div ) * FLANHIP_PI2_F );
				slow |= !( __builtin_fabs( phs[j] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
				}
			}
		const bool any_slow = __any( slow );
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			cf x[4];
			if( any_slow )
				{
				#pragma unroll
				for( int j = 0; j < 4; ++j )
					{
					ph[q][j] = fold_phase_any( ph[q][j] );
					const float2 sc = sincos_wide( float( ph[q][j] ) );
					x[j] = mk( mf[q][j].x * sc.y, mf[q][j].x * sc.x );
					}
				}
			else
				{
				v4f th, m4;
				#pragma unroll
				for( int j = 0; j < 4; ++j ) { ph[q][j] = fold_phase_loop( ph[q][j] ); th[j] = float( ph[q][j] ); m4[j] = mf[q][j].x; }   // :59
				v4f sn, cs;
				sincos_fast_v( th, sn, cs );
				const v4f xr = m4 * cs, xi = m4 * sn;                                // std::polar, :60
				#pragma unroll
				for( int j = 0; j < 4; ++j ) x[j] = cf{ xr[j], xi[j] };
				}
			if( q == 0 )
				{
				// k = 0: X[0], X[2048] are real (c2r ignores their imaginary parts); the quad's second pair is ( 1024, 1024 )
				const bool l0 = k0 == 0;
				x[0].y = l0 ? 0.0f : x[0].y;  x[1].y = l0 ? 0.0f : x[1].y;
				}
			const v4f_t tw = s_twq[64 * q];                                       // ( cos, sin )( 2 pi k / 4096 ), ( cos, -sin )( 2 pi k / 2048 )
			cf zk, zn, zm, zk2;                                                   // Zc[k], Zc[2048-k], Zc[1024-k], Zc[1024+k]
			merge_pair( x[0], x[1], tw.x, tw.y, zk, zn );
			merge_pair( x[2], x[3], tw.y, tw.x, zm, zk2 );                        // exp( +2 pi i ( 1024-k ) / 4096 ) = ( sin, cos )
			const cf dk = mk( zk.x - zk2.x, zk.y - zk2.y ), dm = mk( zm.x - zn.x, zm.y - zn.y );
			const int slot = padl + 68 * ( 4 * role + q ), mslot = mir - 68 * ( 4 * role + q );
			bufA[slot] = mk( zk.x + zk2.x, zk.y + zk2.y );
			bufB[slot] = mk( __builtin_fmaf( dk.x, tw.z, -( dk.y * tw.w ) ), __builtin_fmaf( dk.x, tw.w, dk.y * tw.z ) );       // ( Zc[k] - Zc[k+1024] ) w^k
			bufA[mslot] = mk( zm.x + zn.x, zm.y + zn.y );                          // (k = 0: slot PAD( 1024 ), unused)
			bufB[mslot] = mk( -__builtin_fmaf( dm.x, tw.z, dm.y * tw.w ), __builtin_fmaf( dm.x, tw.w, -( dm.y * tw.z ) ) );    // x w^(1024-k) = x -conj( w^k )
			__builtin_amdgcn_sched_barrier( 0 );                                  // a quad at a time: keeps the temporaries of 16 bins from overlapping
			}
		if( role == 1 )
			{
			// k = 512: bins 512 and 1536 are one pair, w = exp( +i pi / 4 ); A[512] = Zc[512] + Zc[1536], B[512] = ( Zc[512] - Zc[1536] ) ( -i )
			cf x[2];
			#pragma unroll
			for( int j = 0; j < 2; ++j )
				{
				phs[j] = any_slow ? fold_phase_any( phs[j] ) : fold_phase_loop( phs[j] );
				float sn, cs;
				if( any_slow ) { const float2 sc = sincos_wide( float( phs[j] ) ); sn = sc.x; cs = sc.y; } else sincos_fast( float( phs[j] ), sn, cs );
				x[j] = mk( mfs[j].x * cs, mfs[j].x * sn );
				}
			cf zk, zn;
			merge_pair( x[0], x[1], 0.70710678118654752f, 0.70710678118654752f, zk, zn );
			if( lane == 0 )
				{
				bufA[544] = mk( zk.x + zn.x, zk.y + zn.y );                            // slot PAD( 512 )
				bufB[544] = mk( zk.y - zn.y, -( zk.x - zn.x ) );
				}
			}
		};

	if( frames > 0 ) load_row( t0 );
	if( p.group_sums )
		{
		// No scan over the chains ran: `carry` still holds the chains' own sums, group_carry the running phase on entry to every group of TEAMS chains.
		// The running phase on entry to a chain = that, then the chains of this group before it, added and folded in order -- as in k_synthesize_v2, whose words
		// these are: one thread per bin, four or five bins side by side, every load ahead of the dependent additions; every team's carries
		// land in the team's SECOND buffer set, which nobody writes before the first meeting (one set: in the set itself, and the team meets once
		// more before filling it).
		auto fold = []( double r ) { return ( __builtin_fabs( r ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( r ) : fold_phase_any( r ); };
		const int live = min( TEAMS, p.chains_per_channel - group * TEAMS );
		const double * gs = ( p.group_carry ? p.group_carry : p.group_sums ) + int64_t( gchannel ) * gpc * ( N2 + 1 );
		const double * sums0 = p.carry + ( int64_t( gchannel ) * p.chains_per_channel + int64_t( group ) * TEAMS ) * ( N2 + 1 );
		constexpr int NB = ( N2 + NT ) / NT;                                       // bins per thread: 5 for 512 threads (the fifth only for thread 0)
		int bins_of[NB]; bool has[NB]; double run[NB];
		#pragma unroll
		for( int b = 0; b < NB; ++b ) { bins_of[b] = tid + NT * b; has[b] = bins_of[b] <= N2; if( !has[b] ) bins_of[b] = N2; run[b] = 0.0; }
		double vc[NB][TEAMS];
		#pragma unroll
		for( int b = 0; b < NB; ++b )
			{
			#pragma unroll
			for( int w = 0; w < TEAMS; ++w ) vc[b][w] = ( w < live ) ? sums0[int64_t( w ) * ( N2 + 1 ) + bins_of[b]] : 0.0;
			}
		if( p.group_carry )
			{
			#pragma unroll
			for( int b = 0; b < NB; ++b ) run[b] = gs[int64_t( group ) * ( N2 + 1 ) + bins_of[b]];   // the running phase on entry to this group (k_phase_scan2<SEG, true>)
			}
		else
			{
			// few groups per channel: this group adds up the totals of the groups before it itself (see k_synthesize_v2)
			constexpr int BATCH = 8;
			for( int g0 = 0; g0 < group; g0 += BATCH )
				{
				double v[NB][BATCH];
				#pragma unroll
				for( int b = 0; b < NB; ++b )
					{
					#pragma unroll
					for( int u = 0; u < BATCH; ++u ) v[b][u] = ( g0 + u < group ) ? gs[int64_t( g0 + u ) * ( N2 + 1 ) + bins_of[b]] : 0.0;
					}
				#pragma unroll
				for( int u = 0; u < BATCH; ++u )
					{
					#pragma unroll
					for( int b = 0; b < NB; ++b ) run[b] = fold( run[b] + v[b][u] );       // + 0.0 past the end: fold( x ) of a folded x is x
					}
				}
			}
		#pragma unroll
		for( int w = 0; w < TEAMS; ++w )
			{
			#pragma unroll
			for( int b = 0; b < NB; ++b )
				{
				if( has[b] ) reinterpret_cast<double*>( s + L::BUF + w * ( DOUBLE ? 4 : 2 ) * L::BUF_LEN )[team_stage_slot<!DOUBLE>( bins_of[b], 2 * L::BUF_LEN )] = run[b];   // phase_buffer on entry to chain w of the group
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
		const double * mine = reinterpret_cast<const double*>( buf0 );
		auto slot = [&]( int bin ) { return team_stage_slot<!DOUBLE>( bin, 2 * L::BUF_LEN ); };
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			const int k = k0 + 64 * q;
			ph[q][0] = mine[slot( k )]; ph[q][1] = mine[slot( N2 - k )]; ph[q][2] = mine[slot( C - k )]; ph[q][3] = mine[slot( C + k )];
			}
		if( role == 1 ) { phs[0] = mine[slot( 512 )]; phs[1] = mine[slot( 1536 )]; }
		// one buffer set: the stage IS the A / B buffers the first bins_of_row is about to fill -- both wavefronts read their carries first
		if constexpr( !DOUBLE ) team_sync.meet();
		}
	int64_t pos = chain_start;
	int ring_base = 0;                                                          // RING: the ring slot of output position `pos`
	float * const head1 = p.head + chain * p.head_len;
	// count samples from position a0 on leave the ring (slot ring_base on) for the head buffer, the output, or nowhere (outside the output)
	const bool ring_pairs = RING && ( hop & 1 ) == 0 && ( W & 3 ) == 0;           // every position, hop and slot even: pairs stay pairs
	auto ring_emit = [&]( int64_t a0, int count )
		{
		if( ring_pairs )
			{
			// (a0, count, own_start and out_len = F hop are even here: a pair lies on one side of every boundary)
			for( int j = 2 * ( lane + 64 * role ); j < count; j += 256 )
				{
				int i0 = ring_base + j; i0 -= ( i0 >= Wr ) ? Wr : 0;
				cf * slot = reinterpret_cast<cf*>( ring + i0 );
				const cf v = *slot;
				*slot = mk( 0.0f, 0.0f );
				const int64_t a = a0 + j;
				if( a < own_start ) *reinterpret_cast<cf*>( head1 + ( a - chain_start ) ) = v;
				else if( a >= 0 && a < p.out_len ) *reinterpret_cast<cf*>( out1 + a ) = v;
				}
			return;
			}
		for( int j = lane + 64 * role; j < count; j += 128 )
			{
			int i0 = ring_base + j; i0 -= ( i0 >= Wr ) ? Wr : 0;
			const float v = ring[i0];
			ring[i0] = 0.0f;
			const int64_t a = a0 + j;
			if( a < own_start ) head1[a - chain_start] = v;
			else if( a >= 0 && a < p.out_len ) out1[a] = v;
			}
		};
	if( frames > 0 ) bins_of_row( 0 );
	team_sync.meet();
	const int iters = p.L;
	for( int i = 0; i < iters; ++i )
		{
		const int set = DOUBLE ? ( i & 1 ) : 0;
		if( i < frames )
			{
			const int64_t t = t0 + i;
			load_row( min( t + 1, t1 - 1 ) );                                     // (the last frame requests itself again: nobody waits for it)
			const cf * mybuf = buf0 + ( 2 * set + role ) * L::BUF_LEN;
			cf z[16];
			#pragma unroll
			for( int q = 0; q < 16; ++q ) z[q] = mybuf[padl + 68 * q];
			wave_sync();
			fft_fast<10>( z, const_cast<cf*>( mybuf ), s_tw1, s_tw3, lane );
			// G = fft( A or B ): samples 4m (+2 for B) = G[m].x, 4m+1 (+2) = -G[m].y, m = lane + 64 q < 64 WQ; window, overlap-add (AudioPV.cpp:122-134)
			if constexpr( RING )
				{
				// the windowed half frame into the ring: sample 4 m + 2 role (+1) of this frame sits ring_base + that many slots on, modulo Wr
				if( ring_pairs )
					{
					// even hop and window: a sample pair is one aligned 8-byte slot pair that never straddles the ring's end
					#pragma unroll
					for( int q = 0; q < WQ; ++q )
						{
						const int s0 = 4 * ( lane + 64 * q ) + 2 * role;
						if( s0 < W )
							{
							float w0, w1;
							if constexpr( WINGLOB ) { const cf g = *reinterpret_cast<const cf*>( p.window + eo_opaque( s0 ) ); w0 = g.x * p.window_scale; w1 = g.y * p.window_scale; }   // ( W % 4 == 0 here: s0 + 1 < W )
							else { const v4f_t wv = s_win[64 * q]; w0 = role ? wv.z : wv.x; w1 = role ? wv.w : wv.y; }
							int i0 = ring_base + s0; i0 -= ( i0 >= Wr ) ? Wr : 0;
							cf * slot = reinterpret_cast<cf*>( ring + i0 );
							const cf r = *slot;
							*slot = mk( r.x + z[q].x * w0, r.y + ( -z[q].y ) * w1 );
							}
						}
					}
				else
					{
					#pragma unroll
					for( int q = 0; q < WQ; ++q )
						{
						const int s0 = 4 * ( lane + 64 * q ) + 2 * role;
						if( s0 < W )
							{
							float w0, w1 = 0.0f;
							if constexpr( WINGLOB )
								{
								const int so = eo_opaque( s0 );
								w0 = p.window[so] * p.window_scale;
								if( s0 + 1 < W ) w1 = p.window[so + 1] * p.window_scale;
								}
							else { const v4f_t wv = s_win[64 * q]; w0 = role ? wv.z : wv.x; w1 = role ? wv.w : wv.y; }
							int i0 = ring_base + s0; i0 -= ( i0 >= Wr ) ? Wr : 0;
							ring[i0] += z[q].x * w0;
							if( s0 + 1 < W )
								{
								int i1 = i0 + 1; i1 -= ( i1 >= Wr ) ? Wr : 0;
								ring[i1] += ( -z[q].y ) * w1;
								}
							}
						}
					}
				}
			else
			{
			#pragma unroll
			for( int q = 0; q < WQ; ++q )
				{
				const v4f_t wv = s_win[64 * q];                                    // zero beyond W
				acc[q].x += z[q].x * ( role ? wv.z : wv.x );
				acc[q].y += ( -z[q].y ) * ( role ? wv.w : wv.y );
				}
			}
			if constexpr( RING ) {}
			else if constexpr( HS == 0 )
				{
				emit_half( pos, acc[0], 0 );
				const bool low = lane < 32;
				#pragma unroll
				for( int q = 0; q < WQ; ++q )
					{
					const cf nxt = ( q + 1 < WQ ) ? acc[q + 1] : mk( 0.0f, 0.0f );
					acc[q] = mk( rotate_half( acc[q].x, nxt.x, low ), rotate_half( acc[q].y, nxt.y, low ) );
					}
				}
			else
				{
				// HS stores in either arm.  Past the chain's head every step of any chain but a channel's first goes to the output proper, all of
				// it inside: a scalar address (the wavefront's number is a scalar here) plus the lane's constant offset; emit_step for the rest
				if( chain_in_channel != 0 && pos >= own_start )
					{
					#pragma unroll
					for( int q = 0; q < HS; ++q ) *reinterpret_cast<cf*>( reinterpret_cast<char*>( out1 + pos + 256 * q ) + lane_off ) = acc[q];
					}
				else
					{
					#pragma unroll
					for( int q = 0; q < HS; ++q ) emit_step( pos + 256 * q, acc[q] );
					}
				#pragma unroll
				for( int q = 0; q < WQ; ++q ) acc[q] = ( q + HS < WQ ) ? acc[q + HS] : mk( 0.0f, 0.0f );
				}
			if constexpr( !RING )
				{
				pos += hop;
				ov.after_frame( i, frames, pos >= own_start, lane );
				}
			}
		// one buffer set: nobody may write the next frame's A / B before both wavefronts have transformed this one's (the transform uses its
		// buffer as scratch).  Two sets: the next frame goes to the other set
		if constexpr( !DOUBLE ) team_sync.meet();
		if constexpr( RING )
			{
			// both halves of this frame are in the ring: its first hop samples are final -- they leave (4-byte stores: a hop may be odd) and their
			// slots are cleared for the frame that will reach them next (its additions come after the meeting below)
			if( i < frames )
				{
				ring_emit( pos, hop );
				ring_base += hop; ring_base -= ( ring_base >= Wr ) ? Wr : 0;
				pos += hop;
				}
			}
		if( i + 1 < frames ) bins_of_row( DOUBLE ? ( set ^ 1 ) : 0 );
		team_sync.meet();                                                         // the next frame's A / B are written
		}
	if( !active ) return;
	// flush the partial sums that the next chain's head completes; the last chain zero-fills to the end of the output
	const int64_t ring_end = pos + ( W - hop );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	if constexpr( RING )
		{
		// (the loop's last meeting is behind both wavefronts: the ring is complete)
		ring_emit( pos, W - hop );
		for( int64_t a = ring_end + lane + 64 * role; a < flush_end; a += 128 ) if( a >= 0 && a < p.out_len ) out1[a] = 0.0f;
		}
	else if( fix && !last_chain ) {}                                             // (the tail meets the next chain's head below)
	else if constexpr( HS == 0 )
		{
		// W - hop = 1920 (3968) samples = 7.5 (15.5) steps: by halves (the next chain writes from ring_end on itself)
		#pragma unroll
		for( int h = 0; h < 2 * WQ; ++h )
			{
			const int64_t a0 = pos + 128 * h;
			if( a0 < flush_end ) emit_half( a0, acc[h >> 1], h & 1 );
			}
		for( int64_t a0 = pos + 128 * 2 * WQ; a0 < flush_end; a0 += 128 ) emit_half( a0, mk( 0.0f, 0.0f ), 0 );
		}
	else
		{
		#pragma unroll
		for( int q = 0; q < WQ; ++q )
			{
			const int64_t a0 = pos + 256 * q;
			if( a0 < flush_end ) emit_step( a0, acc[q] );
			}
		for( int64_t a0 = pos + 256 * WQ; a0 < flush_end; a0 += 256 ) emit_step( a0, mk( 0.0f, 0.0f ) );
		}
	if constexpr( !RING ) ov.finish( p, acc, 4 * lane + 2 * role, chain, chain_start, pos, out2, lane );
	}

} // namespace flanhip
