// pv_kernels_team.h -- dft 8192 and 16384: a chain walked by a TEAM of R = 4 / 8 wavefronts (one block), each with a 1024-point register transform.
// The dft 4096 team kernels (pv_kernels_eo.h: R = 2) generalised (round 6).
//
// Analysis (Conversions/AudioPV.cpp:12-78).  The frame's packed complex sequence c[n] = x[2n] + i x[2n+1], n < CT = 1024 R (zero from W/2 on), is cut
// by residue:  E_r = FFT1024( c[R m + r] ), one residue per wavefront -- fft_fast<10>, the dft 2048 kernel's register transform, 16 points per lane --
// and joined by one radix-R step through LDS:
//        Z[k + 1024 j] = sum_r  W_R^(r j)  w^(r k) E_r[k],        w = exp( -2 pi i / CT ),  W_R = exp( -2 pi i / R ),  j < R.
// A lane owns k-GROUPS: k = lane + 64 Q, Q < 8, of which a wavefront has KQ = 8 / R.  From E_r[k] and E_r[1024 - k] (every r; 2 R LDS reads) come the
// R values Z[k + 1024 j] and their mirrors Z[CT - k - 1024 j] = DFT_R( conj( w^(r k) ) E_r[1024 - k] )[( R - j ) mod R], i.e. R mirror pairs
// ( b, CT - b ) of the real-transform split: 2 R bins per group, 16 bins per lane and frame -- what a dft 2048 wavefront does, with its register budget
// (16 previous phases, 16 chain sums), at its instruction count per bin plus the join.
//   k = 0 (lane 0 of wavefront 0): the mirror slot PAD( 1024 ) holds a copy of E_r[0], so the group is the general one: pairs ( 1024 j, CT - 1024 j ),
//   each of j = 1 .. R - 1 twice (bit-identical: the twiddle tables are exactly symmetric), j = 0 is DC / Nyquist;
//   k = 512 is its own mirror: the R bins 512 + 1024 j.  Every wavefront leaves its E_r[512] in a ring of 64 frames in LDS, and once per 64 frames (and at
//   the chain's end) wavefront r works off bin 512 + 1024 r of the whole batch, one frame per lane (k_analyze_v2 does the same with its bin C / 2).
// One buffer set: two team meetings per frame (the transform uses its buffer as scratch).  Window and hop are multiples of 128 R samples (a sample pair per
// lane and q-step: q < WQ = W / 128 R).  dft 8192: TWO teams per block share one set of tables and the window table, which is what makes room for the window
// in LDS ([r][q][lane] pairs, up to 8192 samples) -- read through L2 at the end of the bins phase (first version) the window loads sat behind the frame's 16 MF
// stores in the wavefront's memory queue, and the wait for them was a wait for the stores' acknowledgement; dft 16384: one team per block and the
// block barrier; windows above 8192 samples are read from memory.
//
// Synthesis (AudioPV.cpp:86-139): the mirror image.  Per frame a wavefront runs the inverse phase vocoder on its 16 bins, merges the pairs into the
// half-size spectrum Zc (conjugated), and leaves  A_r[k] = w^(r k) DFT_R( Zc[k + 1024 .] )[r]  (and the same at 1024 - k) in buffer r, every r; behind a
// barrier wavefront r transforms A_r: output points g[R m + r], samples 2 ( R m + r ) (+1), windows them and overlap-adds into registers (WQ pairs per lane);
// a finished hop leaves as HS = hop / 128 R stores of 8 bytes per lane.  Bin 512 + 1024 r is wavefront r's every frame (all lanes alike, one bin's
// worth of arithmetic), the R spectrum values meet in LDS and every wavefront works out its own A_r[512].
// Carries: from k_phase_scan2 (the chain sums from the fused analysis or from k_phase_sums2); overlaps of neighbouring chains: k_ola_fixup4.
#pragma once
#include "pv_kernels_eo.h"

namespace flanhip {

struct TeamTables
	{
	const cf * tw1;       // [15][16]      exp( -2 pi i r k / 256 )
	const cf * tw3;       // [3][256]      exp( -2 pi i r j / 1024 )
	const cf * twj;       // [R-1][512]    exp( -2 pi i r k / CT ), r = 1 .. R - 1
	const cf * tws;       // [R/2][512]    exp( -2 pi i ( k + 1024 j ) / 2 CT ), j < R / 2
	const cf * two;       // [2 R]         exp( -2 pi i ( 512 + 1024 j ) / 2 CT ), j < R;  then exp( -2 pi i 512 r / CT ), r < R
	const float * window; // [W]
	};

// WP: sample pairs of the window table in LDS (64 R WQ: the window as [r][q][lane] pairs -- pair R ( lane + 64 q ) + r --, so that a wavefront reads
// consecutive slots), 0: the window is read from memory (windows above 8192 samples)
template<int R, int TEAMS, int WP> struct TeamLds
	{
	static constexpr int S = 1024;
	static constexpr int TW1 = 0;                          // [15][16]
	static constexpr int TW3 = TW1 + 240;                  // [3][256]
	static constexpr int TWJ = TW3 + 768;                  // [R-1][512]
	static constexpr int TWS = TWJ + ( R - 1 ) * 512;      // [R/2][512]  (analysis: halved; synthesis: conjugated)
	static constexpr int WIN = TWS + ( R / 2 ) * 512;      // [WP]
	static constexpr int TEAM0 = WIN + WP;                 // per team:
	static constexpr int BUF_LEN = S + S / 16 + 1;         //   R buffers; slot PAD( 1024 ) = 1088 included
	static constexpr int ORPH = R * BUF_LEN;               //   analysis: [64][R] E_r[512] of the last 64 frames; synthesis: [R] spectrum values of bins 512 + 1024 r
	static constexpr int FLAG = ORPH + 64 * R;             //   the team's meeting counter (TEAMS > 1)
	static constexpr int TEAM_LEN = FLAG + 2;
	static constexpr size_t bytes() { return size_t( TEAM0 + TEAMS * TEAM_LEN ) * 8; }
	};

// The R wavefronts of a team meet.  One team per block: the block barrier.  Several: a counter in the team's LDS, as the dft 4096 kernels' TeamSync
// (pv_kernels_eo.h) -- s_barrier would hold every team of the block to the pace of the slowest wavefront, every frame
template<int R, bool BLOCK> struct TeamMeet
	{
	lds_u32 * flag; unsigned target; int lane;
	__device__ __forceinline__ void meet()
		{
		if constexpr( BLOCK ) lds_block_sync();
		else
			{
			target += R;
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
		}
	};

__device__ __forceinline__ cf cmul_f( cf a, cf w )        // a w, fused like the joins of pv_kernels_eo.h
	{
	return mk( __builtin_fmaf( w.x, a.x, -( w.y * a.y ) ), __builtin_fmaf( w.x, a.y, w.y * a.x ) );
	}
__device__ __forceinline__ cf cmul_fc( cf a, cf w )       // a conj( w )
	{
	return mk( __builtin_fmaf( w.x, a.x, w.y * a.y ), __builtin_fmaf( w.x, a.y, -( w.y * a.x ) ) );
	}
// x exp( -i pi r / R ) = x w16^( 16 r / 2 R )
template<int R, int r> __device__ __forceinline__ cf mul_half_turn( cf a ) { return mul_w16<( 8 / R ) * r>( a ); }

template<int R, int TEAMS, int WQ, bool SUMS, bool WINLDS>
__global__ __launch_bounds__( 64 * R * TEAMS, 2 ) void k_analyze_team( AnalyzeParams p, TeamTables tb )
	{
	using L = TeamLds<R, TEAMS, WINLDS ? 64 * R * WQ : 0>;
	static_assert( R == 4 || R == 8, "teams of four or eight wavefronts" );
	static_assert( WQ == 2 || WQ == 4 || WQ == 8 || WQ == 16, "window / 128 R" );
	constexpr int S = 1024, CT = S * R, KQ = 8 / R, NT = 64 * R * TEAMS, NB = 2 * R;      // NB: bins per k-group
	constexpr int NV = 8, NG = NB / NV;                                            // bins per vector stream, streams per group
	typedef float VB __attribute__(( ext_vector_type( NV ) ));
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 ), team = wave / R, role = wave % R;
	const int W = p.window_size, hop = p.hop;

	// cancellation (core.hip): a team meets in its frame loop, so the decision is the block's
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	for( int i = tid; i < 240; i += NT ) s[L::TW1 + i] = tb.tw1[i];
	for( int i = tid; i < 768; i += NT ) s[L::TW3 + i] = tb.tw3[i];
	for( int i = tid; i < ( R - 1 ) * 512; i += NT ) s[L::TWJ + i] = tb.twj[i];
	for( int i = tid; i < ( R / 2 ) * 512; i += NT ) { const cf a = tb.tws[i]; s[L::TWS + i] = mk( 0.5f * a.x, 0.5f * a.y ); }
	if constexpr( WINLDS )
		{
		// pair n = R m + r, m = lane + 64 q  ->  slot ( r WQ + q ) 64 + lane
		for( int n = tid; n < 64 * R * WQ; n += NT )
			{
			const int r = n % R, m = n / R;
			s[L::WIN + ( r * WQ + ( m >> 6 ) ) * 64 + ( m & 63 )] = *reinterpret_cast<const cf*>( tb.window + 2 * n );
			}
		}
	cf * const buf0 = s + L::TEAM0 + team * L::TEAM_LEN;
	TeamMeet<R, TEAMS == 1> team_sync{ (lds_u32*) reinterpret_cast<unsigned*>( buf0 + L::FLAG ), 0u, lane };
	if( role == 0 && lane == 0 ) *team_sync.flag = 0u;
	__syncthreads();
	if( s_cancel ) return;
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const cf * s_win = s + L::WIN + role * WQ * 64 + lane;
	cf * const mybuf = buf0 + role * L::BUF_LEN;
	cf * const orph = buf0 + L::ORPH;

	// a block is a GROUP: TEAMS consecutive chains of one channel (the last group of a channel may be short: its spare teams retire here)
	const int gpc = ( p.chains_per_channel + TEAMS - 1 ) / TEAMS;
	const int gchannel = int( blockIdx.x ) / gpc, group = int( blockIdx.x ) % gpc;
	const int chain_in_channel = group * TEAMS + team;
	if( chain_in_channel >= p.chains_per_channel ) return;
	const int64_t chain = int64_t( gchannel ) * p.chains_per_channel + chain_in_channel;
	const int channel = gchannel;
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37
	const int padl = lane + ( lane >> 4 );
	const int mir = ( S * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );      // buf[mir - 68 Q] = slot PAD( S - lane - 64 Q )
	const int n32 = int( p.n );
	const float rdft = 1.0f / float( 2 * CT );
	const int k0 = lane + 64 * KQ * role;                                     // k of this wavefront's first group
	const cf * s_twj = s + L::TWJ + k0;
	const cf * s_tws = s + L::TWS + k0;

	float prev[KQ][NB], prevo = 0.0f;
	#pragma unroll
	for( int q = 0; q < KQ; ++q ) { for( int j = 0; j < NB; ++j ) prev[q][j] = 0.0f; }   // AudioPV.cpp:44
	double sm[SUMS ? KQ : 1][NB], smo = 0.0;
	if constexpr( SUMS )
		{
		#pragma unroll
		for( int q = 0; q < KQ; ++q ) { for( int j = 0; j < NB; ++j ) sm[q][j] = 0.0; }
		}
	unsigned mmax = 0u;                                                       // bit patterns, see k_analyze_v2

	const int64_t tfirst = t0 > 0 ? t0 - 1 : t0;
	const int frames = int( t1 - tfirst );                                    // iterations (the halo frame included)
	constexpr int SPAN = 128 * R * WQ;                                        // = W: samples a frame reads
	auto frame_inside = [&]( int64_t t ) { return int64_t( hop ) * t - W / 2 >= 0 && int64_t( hop ) * t - W / 2 + SPAN <= p.n; };
	constexpr std::true_type inside{};
	constexpr std::false_type outside{};

	cf raw[WQ], win[WQ];                                                      // this wavefront's residue of a frame: points lane + 64 q, and their window
	struct __attribute__(( packed, aligned( 4 ) )) f2u { float x, y; };
	auto load_residue = [&]( int64_t t, auto fast_tag )
		{
		constexpr bool FAST = decltype( fast_tag )::value;
		const int start = int( int64_t( hop ) * t - W / 2 ) + 2 * role;
		#pragma unroll
		for( int q = 0; q < WQ; ++q )
			{
			const int a = start + 2 * R * ( lane + 64 * q );
			if constexpr( FAST )
				{
				const f2u v = *reinterpret_cast<const f2u*>( x + a );
				raw[q] = mk( v.x, v.y );
				}
			else
				{
				const float v0 = ( a >= 0 && a < n32 ) ? x[a] : 0.0f;                 // zero outside the signal (AudioPV.cpp:54-62)
				const float v1 = ( a + 1 >= 0 && a + 1 < n32 ) ? x[a + 1] : 0.0f;
				raw[q] = mk( v0, v1 );
				}
			}
		};
	auto load_window = [&]()
		{
		if constexpr( !WINLDS )
			{
			#pragma unroll
			for( int q = 0; q < WQ; ++q ) win[q] = *reinterpret_cast<const cf*>( tb.window + eo_opaque( 2 * R * ( lane + 64 * q ) + 2 * role ) );
			}
		};
	auto transform_frame = [&]( int fi )
		{
		cf z[16];
		#pragma unroll
		for( int q = 0; q < WQ; ++q )
			{
			const cf wv = WINLDS ? s_win[64 * q] : win[q];
			z[q] = mk( raw[q].x * wv.x, raw[q].y * wv.y );                         // AudioPV.cpp:65
			}
		#pragma unroll
		for( int q = WQ; q < 16; ++q ) z[q] = mk( 0.0f, 0.0f );
		// (the lane number through an empty asm: the transform's lane-derived LDS addresses are loop invariants the compiler otherwise keeps -- and, at 256
		// registers, spills: a scratch reload in here is a wait for the frame's MF stores)
		const int lo = eo_opaque( lane );
		fft_fast<10>( z, mybuf, s_tw1, s_tw3, lo );
		const int padl_o = lo + ( lo >> 4 );
		#pragma unroll
		for( int q = 0; q < 16; ++q ) mybuf[padl_o + 68 * q] = z[q];               // natural order: slot PAD( lane + 64 q )
		if( lane == 0 )
			{
			mybuf[1088] = z[0];                                                    // E_r[1024] = E_r[0]: the k = 0 group's mirror
			orph[( fi & 63 ) * R + role] = z[8];                                   // E_r[512]
			}
		};

	auto bins_of_frame = [&]( int64_t t, int64_t tn, auto halo_tag, auto next_fast )
		{
		constexpr bool halo = decltype( halo_tag )::value;
		cf * row = reinterpret_cast<cf*>( p.out + ( int64_t( channel ) * p.F + t ) * ( CT + 1 ) );
		cf * row_a = row + k0;                                                  // bin k + 1024 j        (+ 64 q)
		cf * row_b = row + ( CT - k0 );                                         // bin CT - k - 1024 j   (- 64 q)
		#pragma unroll
		for( int q = 0; q < KQ; ++q )
			{
			if( q == KQ / 2 ) load_residue( tn, next_fast );                      // the next frame's samples travel during the (second half of the) bins
			const int Q = KQ * role + q;                                           // (scalar)
			cf T[R], V[R];
			#pragma unroll
			for( int r = 0; r < R; ++r )
				{
				T[r] = buf0[r * L::BUF_LEN + padl + 68 * Q];
				V[r] = buf0[r * L::BUF_LEN + mir - 68 * Q];
				}
			#pragma unroll
			for( int r = 1; r < R; ++r )
				{
				const cf w = s_twj[( r - 1 ) * 512 + 64 * q];
				T[r] = cmul_f( T[r], w );
				V[r] = cmul_fc( V[r], w );
				}
			dft_reg<R>( T );
			dft_reg<R>( V );
			float re[NB], im[NB];
			#pragma unroll
			for( int j = 0; j < R; ++j )
				{
				cf w = s_tws[( j % ( R / 2 ) ) * 512 + 64 * q];
				if( j >= R / 2 ) w = mk( w.y, -w.x );                               // x exp( -i pi / 2 )
				split_pair( T[j], V[( R - j ) % R], w.x, w.y, re[2 * j], im[2 * j], re[2 * j + 1], im[2 * j + 1] );
				}
			if( q == 0 )
				{
				const bool l0 = k0 == 0;                                            // Z[0] -> X[0], X[CT]
				re[0] = l0 ? T[0].x + T[0].y : re[0];  im[0] = l0 ? 0.0f : im[0];
				re[1] = l0 ? T[0].x - T[0].y : re[1];  im[1] = l0 ? 0.0f : im[1];
				}
			const float fk = float( k0 + 64 * q );
			#pragma unroll
			for( int g = 0; g < NG; ++g )
				{
				VB vre, vim, pv, binf;
				#pragma unroll
				for( int i = 0; i < NV; ++i )
					{
					const int e = NV * g + i, j = e >> 1;
					vre[i] = re[e]; vim[i] = im[e]; pv[i] = prev[q][e];
					binf[i] = ( ( e & 1 ) ? float( CT - S * j ) - fk : float( S * j ) + fk ) * p.sample_rate * rdft;      // PVBuffer.cpp:443-446
					}
				VB phase, m;
				polar_v( vre, vim, phase, m );                                      // phase_vocoder.cpp:37-52 (AudioPV.cpp:69-73)
				#pragma unroll
				for( int i = 0; i < NV; ++i ) prev[q][NV * g + i] = phase[i];       // :45
				if constexpr( !halo )
					{
					const VB expd = div_c_each( binf, p.ar_div ) * vsplat<VB>( FLANHIP_PI2_F );       // :47
					const VB delta_phase = ( phase - pv ) - expd;                                // :44, :47-48
					VB wrapped = delta_phase;
					if( use_wrapping ) wrapped = delta_phase - vsplat<VB>( FLANHIP_PI2_F ) * round_half_away_v( div_pi2_v( delta_phase ) );   // :39-42,49
					VB war;
					#pragma unroll
					for( int i = 0; i < NV; ++i ) war[i] = wrapped[i] * p.analysis_rate;
					const VB f = binf + div_pi2_v( war );                                    // :50-52
					#pragma unroll
					for( int i = 0; i < NV; ++i )
						{
						const int e = NV * g + i, j = e >> 1;
						if( e & 1 ) __builtin_nontemporal_store( cf{ m[i], f[i] }, row_b - S * j - 64 * q );
						else __builtin_nontemporal_store( cf{ m[i], f[i] }, row_a + S * j + 64 * q );
						}
					if constexpr( SUMS )
						{
						const VB term = div_c_each( f, p.ar_div ) * vsplat<VB>( FLANHIP_PI2_F );      // phase_vocoder.cpp:57-58
						#pragma unroll
						for( int i = 0; i < NV; ++i ) sm[q][NV * g + i] += double( term[i] );
						#pragma unroll
						for( int i = 0; i < NV; i += 2 ) mmax = max( mmax, max( __float_as_uint( m[i] ), __float_as_uint( m[i + 1] ) ) );
						}
					}
				}
			}
		};

	// bin 512 + 1024 role of the frames tb .. tb + nb - 1 (lane j: frame tb + j; frame tfirst, the halo, only lends its phase)
	auto flush_orphans = [&]( int64_t tb0, int nb )
		{
		const int64_t t = tb0 + lane;
		const bool valid = lane < nb && t >= t0;
		cf T[R];
		#pragma unroll
		for( int r = 0; r < R; ++r ) T[r] = orph[lane * R + r];
		// T_r = w^( 512 r ) E_r[512] = exp( -i pi r / R ) E_r[512]
		if constexpr( R == 4 ) { T[1] = mul_half_turn<4, 1>( T[1] ); T[2] = mul_half_turn<4, 2>( T[2] ); T[3] = mul_half_turn<4, 3>( T[3] ); }
		else
			{
			T[1] = mul_half_turn<8, 1>( T[1] ); T[2] = mul_half_turn<8, 2>( T[2] ); T[3] = mul_half_turn<8, 3>( T[3] ); T[4] = mul_half_turn<8, 4>( T[4] );
			T[5] = mul_half_turn<8, 5>( T[5] ); T[6] = mul_half_turn<8, 6>( T[6] ); T[7] = mul_half_turn<8, 7>( T[7] );
			}
		dft_reg<R>( T );                                                         // T[j] = Z[512 + 1024 j]
		cf zk = T[0], zm = T[R - 1];
		#pragma unroll
		for( int j = 1; j < R; ++j ) { zk = ( role == j ) ? T[j] : zk; zm = ( role == j ) ? T[R - 1 - j] : zm; }
		const cf w = tb.two[role];
		float rk, ik, rm, im;
		split_pair( zk, zm, 0.5f * w.x, 0.5f * w.y, rk, ik, rm, im );
		const int bin = 512 + S * role;
		cf * rowp = reinterpret_cast<cf*>( p.out + ( int64_t( channel ) * p.F + ( valid ? t : t0 ) ) * ( CT + 1 ) );
		const int j0 = __builtin_amdgcn_readfirstlane( ( tb0 < t0 ) ? 1 : 0 ), j1 = __builtin_amdgcn_readfirstlane( nb );
		const float phase = atan2_fast( ik, rk );
		float pvx = __shfl_up( phase, 1 );                                     // the frame before: the lane below, or the batch before
		pvx = ( lane == 0 ) ? prevo : pvx;
		prevo = __shfl( phase, nb - 1 );
		const float bx = float( bin ) * p.sample_rate * rdft;
		const float delta_phase = ( phase - pvx ) - div_c( bx, p.ar_div ) * FLANHIP_PI2_F;
		const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * round_half_away( div_pi2( delta_phase ) ) : delta_phase;
		const float f = bx + div_pi2( wrapped * p.analysis_rate );
		const float m = magnitude_scaled( rk, ik );
		if( valid ) __builtin_nontemporal_store( mk( m, f ), rowp + bin );
		if constexpr( SUMS )
			{
			const float term = div_c( f, p.ar_div ) * FLANHIP_PI2_F;             // phase_vocoder.cpp:57-58
			for( int l = j0; l < j1; ++l ) smo += double( __uint_as_float( __builtin_amdgcn_readlane( __float_as_uint( term ), l ) ) );   // frame order
			mmax = valid ? max( mmax, __float_as_uint( m ) ) : mmax;
			}
		};

	// iteration i of the chain: frame tfirst + i (the first is the halo when t0 > 0).  Rotated like the dft 2048 kernel: the bins of frame i, then the
	// transform of frame i + 1; a barrier after each
	load_window();
	if( frame_inside( tfirst ) ) load_residue( tfirst, inside ); else load_residue( tfirst, outside );
	transform_frame( 0 );
	team_sync.meet();
	auto iteration = [&]( int i, auto halo_tag )
		{
		const int64_t t = tfirst + i, tn = min( t + 1, t1 - 1 );               // (the last frame requests itself again: nobody waits for it)
		if( frame_inside( tn ) ) bins_of_frame( t, tn, halo_tag, inside ); else bins_of_frame( t, tn, halo_tag, outside );
		if( ( i & 63 ) == 63 || i == frames - 1 ) flush_orphans( t - ( i & 63 ), ( i & 63 ) + 1 );
		load_window();                                                          // (under the barrier: the bins' temporaries are dead)
		team_sync.meet();                                                       // nobody writes the next frame's E_r before everybody has read this one's
		if( i + 1 < frames ) transform_frame( i + 1 );
		team_sync.meet();                                                       // the next frame's E_r are written
		};
	if( t0 > 0 ) iteration( 0, std::true_type{} ); else iteration( 0, std::false_type{} );
	for( int i = 1; i < frames; ++i ) iteration( i, std::false_type{} );

	if constexpr( SUMS )
		{
		// the chain's sums, folded like phase_vocoder.cpp:59: what k_phase_sums2 would leave in the workspace
		bool bad = mmax >= 0x7f800000u;
		auto fold = [&]( double sq ) -> double
			{
			bad |= !( __builtin_fabs( sq ) <= 1.7976931348623157e308 );              // a NaN / Inf frequency poisons its sum
			return ( __builtin_fabs( sq ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( sq ) : fold_phase_any( sq );
			};
		double * dst = p.sums + chain * ( CT + 1 );
		#pragma unroll
		for( int q = 0; q < KQ; ++q )
			{
			const int k = k0 + 64 * q;
			#pragma unroll
			for( int e = 0; e < NB; ++e )
				{
				const int j = e >> 1;
				dst[( e & 1 ) ? CT - S * j - k : S * j + k] = fold( sm[q][e] );        // (k = 0: bins 1024 j twice, the same value)
				}
			}
		const double so = fold( smo );
		if( lane == 0 ) dst[512 + S * role] = so;
		const bool any_bad = __any( bad );
		if( p.nan_out && lane == 0 )
			{
			if( chain == 0 && role == 0 ) { p.nan_out[2] = p.nan_epoch; p.nan_out[4] = p.nan_epoch; }
			if( any_bad ) p.nan_out[0] = p.nan_epoch;
			}
		}
	}


// =================================================================================================================
// PV::convert_to_audio (Conversions/AudioPV.cpp:86-139), dft 8192 / 16384: see the head of the file.  HS = hop / 128 R (0: hop = 64 R, half a step; -4 / -8: a quarter / an eighth), WQ = W / 128 R.
template<int R, int TEAMS, int WQ, int HS, bool WINLDS>
__global__ __launch_bounds__( 64 * R * TEAMS, 2 ) void k_synthesize_team( SynthParams p, TeamTables tb )
	{
	using L = TeamLds<R, TEAMS, WINLDS ? 64 * R * WQ : 0>;
	static_assert( R == 4 || R == 8, "teams of four or eight wavefronts" );
	static_assert( HS <= WQ && ( HS >= 0 || HS == -4 || HS == -8 ), "hop <= window; HS = 0: half a step, -4 / -8: a quarter / an eighth" );
	constexpr int S = 1024, CT = S * R, KQ = 8 / R, NT = 64 * R * TEAMS, NB = 2 * R, STEP = 128 * R;
	constexpr int FR = HS > 0 ? 1 : HS == 0 ? 2 : -HS;                          // a hop is 1 / FR of a step below a whole step: LN = 64 / FR lanes of an accumulator register
	constexpr int LN = 64 / FR;
	constexpr int hop = HS > 0 ? HS * STEP : STEP / FR, W = WQ * STEP;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 ), team = wave / R, role = wave % R;

	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	for( int i = tid; i < 240; i += NT ) s[L::TW1 + i] = tb.tw1[i];
	for( int i = tid; i < 768; i += NT ) s[L::TW3 + i] = tb.tw3[i];
	for( int i = tid; i < ( R - 1 ) * 512; i += NT ) s[L::TWJ + i] = tb.twj[i];
	for( int i = tid; i < ( R / 2 ) * 512; i += NT ) { const cf a = tb.tws[i]; s[L::TWS + i] = mk( a.x, -a.y ); }     // exp( +2 pi i b / 2 CT )
	if constexpr( WINLDS )
		{
		for( int n = tid; n < 64 * R * WQ; n += NT )                            // AudioPV.cpp:102; the analysis kernel's layout
			{
			const int r = n % R, m = n / R;
			const cf g = *reinterpret_cast<const cf*>( tb.window + 2 * n );
			s[L::WIN + ( r * WQ + ( m >> 6 ) ) * 64 + ( m & 63 )] = mk( g.x * p.window_scale, g.y * p.window_scale );
			}
		}
	cf * const buf0 = s + L::TEAM0 + team * L::TEAM_LEN;
	TeamMeet<R, TEAMS == 1> team_sync{ (lds_u32*) reinterpret_cast<unsigned*>( buf0 + L::FLAG ), 0u, lane };
	if( role == 0 && lane == 0 ) *team_sync.flag = 0u;
	__syncthreads();
	if( s_cancel ) return;
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const cf * s_win = s + L::WIN + role * WQ * 64 + lane;
	cf * const mybuf = buf0 + role * L::BUF_LEN;
	cf * const xorph = buf0 + L::ORPH;

	const int gpc = ( p.chains_per_channel + TEAMS - 1 ) / TEAMS;
	const int gchannel = int( blockIdx.x ) / gpc, group = int( blockIdx.x ) % gpc;
	const int chain_in_channel = group * TEAMS + team;
	if( chain_in_channel >= p.chains_per_channel ) return;
	const int64_t chain = int64_t( gchannel ) * p.chains_per_channel + chain_in_channel;
	const int channel = gchannel;
	const int64_t t0 = int64_t( chain_in_channel ) * p.L;
	const int64_t t1 = min( t0 + int64_t( p.L ), p.F );
	const bool last_chain = chain_in_channel == p.chains_per_channel - 1;
	const int frames = int( t1 - t0 );
	float * out1 = p.out + int64_t( channel ) * p.out_len;
	cf * out2 = reinterpret_cast<cf*>( out1 );
	cf * head2 = reinterpret_cast<cf*>( p.head + chain * p.head_len );
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;
	const int padl = lane + ( lane >> 4 );
	const int mir = ( S * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );
	const int k0 = lane + 64 * KQ * role;
	const cf * s_twj = s + L::TWJ + k0;
	const cf * s_tws = s + L::TWS + k0;
	const int obin = 512 + S * role;                                            // this wavefront's bin of the k = 512 group

	// running phases (AudioPV.cpp:105) on entry to the chain: [q][2 j] = bin k + 1024 j, [q][2 j + 1] = bin CT - k - 1024 j
	double ph[KQ][NB], pho;
		{
		const double * carry = p.carry + chain * ( CT + 1 );
		#pragma unroll
		for( int q = 0; q < KQ; ++q )
			{
			const int k = k0 + 64 * q;
			#pragma unroll
			for( int e = 0; e < NB; ++e ) ph[q][e] = carry[( e & 1 ) ? CT - S * ( e >> 1 ) - k : S * ( e >> 1 ) + k];
			}
		pho = carry[obin];
		}
	cf acc[WQ];                                                                 // acc[q] <-> samples pos + STEP q + 2 R lane + 2 role (+1)
	#pragma unroll
	for( int q = 0; q < WQ; ++q ) acc[q] = mk( 0.0f, 0.0f );

	// the chains' overlaps added here (ChainOverlap, pv_kernels_eo.h): a word per wavefront of a chain, where the workspace holds them (p.fix_state)
	ChainOverlap<WQ, STEP> ov;
	ov.init( p, true, chain, R, role, chain_in_channel == 0, last_chain );
	const bool fix = ov.on;
	cf * dump2 = reinterpret_cast<cf*>( p.dump ) + lane;
	auto emit_step = [&]( int64_t a0, cf v )
		{
		const int64_t a = a0 + 2 * R * lane + 2 * role;
		cf * dst = ( a0 < own_start ) ? head2 + ( ( a - chain_start ) >> 1 ) : out2 + ( a >> 1 );
		if( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) dst = dump2;
		if( fix && a0 < own_start ) st_agent( dst, v );                          // (the head another wavefront may come to add up)
		else *dst = v;
		};
	// HS = 0 (hop = half a step: the reference API's hop = window / 16 at window = dft / 2 -- ( 4096, 256, 8192 ), ( 8192, 512, 16384 )): a finished hop is the lower 32
	// lanes of acc[0], and the accumulator moves on by 32 lanes (k_synthesize_eo_team's hop 128: one v_permlane32_swap and one select per register).  The lanes
	// 32 part .. 32 part + 31 of v hold samples a0 + 2 R ( lane & 31 ) + 2 role (+1); the other lanes dump (never at agent scope: pv_kernels_eo.h)
	// (HS = -4 / -8, hop = a quarter / an eighth of a step -- the reference API's default hop 128 kept while dft and window grow: ( 2048, 128, 8192 ),
	// ( 2048, 128, 16384 ): the same with 16 / 8 lanes per hop, the accumulator moving on through ds_bpermute)
	auto emit_half = [&]( int64_t a0, cf v, int part )
		{
		const int lp = lane - LN * part;                                           // the lane's place in the part
		const int64_t a = a0 + 2 * R * lp + 2 * role;
		cf * dst = ( a0 < own_start ) ? head2 + ( ( a - chain_start ) >> 1 ) : out2 + ( a >> 1 );
		const bool idle = lp < 0 || lp >= LN;
		if( idle || ( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) ) dst = dump2;
		if( fix && a0 < own_start ) { if( !idle ) st_agent( dst, v ); }
		else *dst = v;
		};
	auto rotate_half = []( float a, float b, bool low ) -> float          // lanes 0..31 <- a's lanes 32..63, lanes 32..63 <- b's lanes 0..31
		{
		const auto r = __builtin_amdgcn_permlane32_swap( __float_as_uint( a ), __float_as_uint( b ), false, false );
		return __uint_as_float( low ? r[1] : r[0] );
		};
	cf mf[KQ][NB], mfo;
	auto load_row = [&]( int64_t t )
		{
		const cf * row = reinterpret_cast<const cf*>( p.pv + ( int64_t( channel ) * p.F + t ) * ( CT + 1 ) );
		const cf * ra = row + k0, * rb = row + ( CT - k0 );
		#pragma unroll
		for( int q = 0; q < KQ; ++q )
			{
			#pragma unroll
			for( int e = 0; e < NB; ++e ) mf[q][e] = __builtin_nontemporal_load( ( e & 1 ) ? rb - S * ( e >> 1 ) - 64 * q : ra + S * ( e >> 1 ) + 64 * q );
			}
		mfo = __builtin_nontemporal_load( row + obin );
		};
	cf win[WQ];
	auto load_window = [&]()
		{
		if constexpr( !WINLDS )
			{
			#pragma unroll
			for( int q = 0; q < WQ; ++q )
				{
				const cf g = *reinterpret_cast<const cf*>( tb.window + eo_opaque( 2 * R * ( lane + 64 * q ) + 2 * role ) );
				win[q] = mk( g.x * p.window_scale, g.y * p.window_scale );
				}
			}
		};
	// Zc[j], Zc[N-j] from X[j] = a, X[N-j] = b and w = exp( +2 pi i j / 2N ) (k_synthesize_v2's merge)
	auto merge_pair = []( cf a, cf b, float wx, float wy, cf & zj, cf & zn )
		{
		const float ax = a.x + b.x, ay = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
		const float bx = __builtin_fmaf( wx, dx, -( wy * dy ) ), by = __builtin_fmaf( wx, dy, wy * dx );
		zj = mk( ax - by, -( ay + bx ) );
		zn = mk( ax + by, ay - bx );
		};
	// inverse phase vocoder of the row in mf[] (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61), merge, A_r into the buffers
	auto bins_of_row = [&]()
		{
		bool slow = false;
		float dv[KQ][NB], dvo;
		if( p.ar_div.exact )
			{
			const float dc = p.ar_div.c, drc = p.ar_div.rc;
			#pragma unroll
			for( int q = 0; q < KQ; ++q )
				{
				#pragma unroll
				for( int e = 0; e < NB; ++e ) { const float xx = mf[q][e].y, q0 = xx * drc; dv[q][e] = __builtin_fmaf( __builtin_fmaf( -q0, dc, xx ), drc, q0 ); }   // pv_math.h: div_c
				}
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < KQ; ++q )
				{
				#pragma unroll
				for( int e = 0; e < NB; ++e ) dv[q][e] = mf[q][e].y / p.ar_div.c;
				}
			}
		dvo = div_c( mfo.y, p.ar_div );
		#pragma unroll
		for( int q = 0; q < KQ; ++q )
			{
			#pragma unroll
			for( int e = 0; e < NB; ++e )
				{
				ph[q][e] += double( dv[q][e] * FLANHIP_PI2_F );                        // phase_vocoder.cpp:57-58
				slow |= !( __builtin_fabs( ph[q][e] ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
				}
			}
		pho += double( dvo * FLANHIP_PI2_F );
		slow |= !( __builtin_fabs( pho ) < double( FLANHIP_SINCOS_FAST_LIMIT ) );
		const bool any_slow = __any( slow );
		#pragma unroll
		for( int q = 0; q < KQ; ++q )
			{
			cf X[NB];
			#pragma unroll
			for( int g = 0; g < NB / 4; ++g )
				{
				if( any_slow )
					{
					#pragma unroll
					for( int i = 0; i < 4; ++i )
						{
						const int e = 4 * g + i;
						ph[q][e] = fold_phase_any( ph[q][e] );
						const float2 sc = sincos_wide( float( ph[q][e] ) );
						X[e] = mk( mf[q][e].x * sc.y, mf[q][e].x * sc.x );
						}
					}
				else
					{
					v4f th, m4;
					#pragma unroll
					for( int i = 0; i < 4; ++i ) { const int e = 4 * g + i; ph[q][e] = fold_phase_loop( ph[q][e] ); th[i] = float( ph[q][e] ); m4[i] = mf[q][e].x; }   // :59
					v4f sn, cs;
					sincos_fast_v( th, sn, cs );
					const v4f xr = m4 * cs, xi = m4 * sn;                                // std::polar, :60
					#pragma unroll
					for( int i = 0; i < 4; ++i ) X[4 * g + i] = cf{ xr[i], xi[i] };
					}
				}
			if( q == 0 )
				{
				// k = 0: X[0], X[CT] are real (c2r ignores their imaginary parts)
				const bool l0 = k0 == 0;
				X[0].y = l0 ? 0.0f : X[0].y;  X[1].y = l0 ? 0.0f : X[1].y;
				}
			cf D[R], Y[R];                                                         // D[j] = Zc[k + 1024 j], Y[j] = Zc[CT - k - 1024 ( ( R - j ) mod R )]
			#pragma unroll
			for( int j = 0; j < R; ++j )
				{
				cf w = s_tws[( j % ( R / 2 ) ) * 512 + 64 * q];
				if( j >= R / 2 ) w = mk( -w.y, w.x );                               // x exp( +i pi / 2 )
				merge_pair( X[2 * j], X[2 * j + 1], w.x, w.y, D[j], Y[( R - j ) % R] );
				}
			dft_reg<R>( D );
			dft_reg<R>( Y );
			const int Q = KQ * role + q;
			buf0[padl + 68 * Q] = D[0];
			buf0[mir - 68 * Q] = Y[0];                                              // (k = 0: slot PAD( 1024 ), unused)
			#pragma unroll
			for( int r = 1; r < R; ++r )
				{
				const cf w = s_twj[( r - 1 ) * 512 + 64 * q];
				buf0[r * L::BUF_LEN + padl + 68 * Q] = cmul_f( D[r], w );
				buf0[r * L::BUF_LEN + mir - 68 * Q] = cmul_fc( Y[r], w );
				}
			__builtin_amdgcn_sched_barrier( 0 );                                  // a group at a time: keeps the temporaries of 16 bins from overlapping
			}
			{
			// bin 512 + 1024 role: every lane alike
			pho = any_slow ? fold_phase_any( pho ) : fold_phase_loop( pho );
			float sn, cs;
			if( any_slow ) { const float2 sc = sincos_wide( float( pho ) ); sn = sc.x; cs = sc.y; } else sincos_fast( float( pho ), sn, cs );
			if( lane == 0 ) xorph[role] = mk( mfo.x * cs, mfo.x * sn );
			}
		};
	// A_role[512] from the R spectrum values of the k = 512 group (pairs ( j, R - 1 - j ), w = exp( +i pi ( 2 j + 1 ) / 2 R ))
	auto orphan_point = [&]() -> cf
		{
		cf Xo[R], Zo[R];
		#pragma unroll
		for( int j = 0; j < R; ++j ) Xo[j] = xorph[j];
		#pragma unroll
		for( int j = 0; j < R / 2; ++j )
			{
			const cf w = tb.two[j];
			merge_pair( Xo[j], Xo[R - 1 - j], w.x, -w.y, Zo[j], Zo[R - 1 - j] );
			}
		dft_reg<R>( Zo );
		cf d = Zo[0];
		#pragma unroll
		for( int j = 1; j < R; ++j ) d = ( role == j ) ? Zo[j] : d;
		return cmul_f( d, tb.two[R + role] );                                   // x w^( 512 role )
		};

	load_row( t0 );
	bins_of_row();
	load_window();
	team_sync.meet();
	int64_t pos = chain_start;
	for( int i = 0; i < frames; ++i )
		{
		const int64_t t = t0 + i;
		load_row( min( t + 1, t1 - 1 ) );                                       // (the last frame requests itself again: nobody waits for it)
		cf z[16];
		#pragma unroll
		for( int q = 0; q < 16; ++q ) z[q] = mybuf[padl + 68 * q];
		const cf a512 = orphan_point();
		z[8] = ( lane == 0 ) ? a512 : z[8];
		wave_sync();
		fft_fast<10>( z, mybuf, s_tw1, s_tw3, lane );
		// g[R m + role] = z[q], m = lane + 64 q: samples 2 ( R m + role ) = z.x, + 1 = -z.y; window, overlap-add (AudioPV.cpp:122-134)
		#pragma unroll
		for( int q = 0; q < WQ; ++q )
			{
			const cf wv = WINLDS ? s_win[64 * q] : win[q];
			acc[q].x += z[q].x * wv.x;
			acc[q].y += ( -z[q].y ) * wv.y;
			}
		if constexpr( HS == 0 )
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
		else if constexpr( HS < 0 )
			{
			// the accumulator moves on by LN lanes: new acc[q] = { lanes LN .. 63 of acc[q], lanes 0 .. LN - 1 of acc[q + 1] }
			emit_half( pos, acc[0], 0 );
			const int from = ( ( lane + LN ) & 63 ) * 4;
			const bool keep = lane < 64 - LN;
			cf rot = mk( __int_as_float( __builtin_amdgcn_ds_bpermute( from, __float_as_int( acc[0].x ) ) ), __int_as_float( __builtin_amdgcn_ds_bpermute( from, __float_as_int( acc[0].y ) ) ) );
			#pragma unroll
			for( int q = 0; q < WQ; ++q )
				{
				cf nxt = mk( 0.0f, 0.0f );
				if( q + 1 < WQ ) nxt = mk( __int_as_float( __builtin_amdgcn_ds_bpermute( from, __float_as_int( acc[q + 1].x ) ) ), __int_as_float( __builtin_amdgcn_ds_bpermute( from, __float_as_int( acc[q + 1].y ) ) ) );
				acc[q] = mk( keep ? rot.x : nxt.x, keep ? rot.y : nxt.y );
				rot = nxt;
				}
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < HS; ++q ) emit_step( pos + STEP * q, acc[q] );
			#pragma unroll
			for( int q = 0; q < WQ; ++q ) acc[q] = ( q + HS < WQ ) ? acc[q + HS] : mk( 0.0f, 0.0f );
			}
		pos += hop;
		ov.after_frame( i, frames, pos >= own_start, lane );
		team_sync.meet();                                                       // nobody writes the next frame's A_r before everybody has transformed this one's
		if( i + 1 < frames ) { bins_of_row(); load_window(); }
		team_sync.meet();                                                       // the next frame's A_r are written
		}
	// flush the partial sums that the next chain's head completes; the last chain zero-fills to the end of the output
	const int64_t ring_end = pos + ( W - hop );
	const int64_t flush_end = last_chain ? max( ring_end, p.out_len ) : ring_end;
	if( !fix || last_chain )
		{
		if constexpr( HS <= 0 )
			{
			// W - hop = WQ - 1 / FR steps: by parts of a step (the next chain writes from ring_end on itself)
			#pragma unroll
			for( int h = 0; h < FR * WQ; ++h )
				{
				const int64_t a0 = pos + ( STEP / FR ) * h;
				if( a0 < flush_end ) emit_half( a0, acc[h / FR], h % FR );
				}
			for( int64_t a0 = pos + STEP * WQ; a0 < flush_end; a0 += STEP / FR ) emit_half( a0, mk( 0.0f, 0.0f ), 0 );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < WQ; ++q )
				{
				const int64_t a0 = pos + STEP * q;
				if( a0 < flush_end ) emit_step( a0, acc[q] );
				}
			for( int64_t a0 = pos + STEP * WQ; a0 < flush_end; a0 += STEP ) emit_step( a0, mk( 0.0f, 0.0f ) );
			}
		}
	ov.finish( p, acc, 2 * R * lane + 2 * role, chain, chain_start, pos, out2, lane );
	}

} // namespace flanhip
