// pv_kernels_v2.h -- second generation of the dft 2048 analysis / synthesis kernels (C = 1024 complex points per frame).
//
// Same chain decomposition and the same register-resident 16 x 16 x 4 transform as pv_kernels_fast.h (fft_fast), 8-wave blocks = 2
// wavefronts per SIMD at <= 256 VGPRs (12-wave blocks at <= 168 VGPRs were measured in round 2 and dropped: no faster).  The kernels are
// bound by the SIMD's issue / execute time, not by memory or latency (profiles/r03_stall_attribution.txt): what this file does about it
// is fewer and cheaper instructions per frame:
//   * a lane owns bin PAIRS ( k, C - k ), k = lane + 64 q, q < 8: the real-transform split of both bins shares its sums,
//     differences and twiddle products (half the work of splitting each bin on its own), only the upper half of the spectrum
//     crosses LDS for the mirror exchange, and lane 0's pair ( 0, C ) is DC / Nyquist; bin C/2 is the one bin left over;
//   * no register copy of the next frame: a frame's samples are loaded INTO the registers of the spectrum as those die (the upper
//     half right after the mirror exchange, z[q] as soon as pair q has been split), a whole frame ahead of their use;
//   * chains whose frames all lie inside the signal (all but the first and last chain of a channel) take loads with a
//     wave-uniform base and immediate offsets; the clamped / patched loads of the edge frames live in a second loop body;
//   * per-bin constants (bin frequency, expected phase advance) come from an LDS table built in the prologue;
//   * |z| = max * sqrt( 1 + q^2 ) with the q = min / max that atan2 needs anyway (no exponent juggling, never overflows);
//   * roundf( x ) = trunc( x + copysign( 0.49999997, x ) ) (exhaustively equal, tools/check_round_trick.c);
//   * the NaN / Inf scan rides on the fp64 sums (a non-finite f makes its sum non-finite) and a running maximum of m.
#pragma once
#include "pv_kernels_fast.h"

namespace flanhip {

struct V2Lds
	{
	static constexpr int C = 1024;
	static constexpr int TW1 = 0;                          // [15][16]
	static constexpr int TW3 = TW1 + 240;                  // [3][256]
	static constexpr int W2H = TW3 + 768;                  // [512]   0.5 exp(-2 pi i k / 2C), k < C/2
	static constexpr int WIN = W2H + 512;                  // [2048 floats]
	static constexpr int KC = WIN + 1024;                  // analysis: [512] float4 { binf(k), expected(k), binf(C-k), expected(C-k) }
	static constexpr int BUF = KC + 1024;
	static constexpr int BUF_LEN = C + C / 16 + 1;
	static constexpr size_t bytes( int waves ) { return size_t( BUF + waves * BUF_LEN ) * 8; }
	};


#ifdef FLANHIP_STAMPS
// Diagnostic build only (tools/scripts/build_diag.sh stamps): where does a frame spend its cycles?  s_memtime between the sections of the
// frame loop, per-section sums in scalar registers, added to g_stamp_acc once per wavefront.  Never quote this build's run time.
__device__ unsigned long long g_stamp_acc[16];
__device__ unsigned long long g_stamp_span[2 * 4096];      // per wavefront (block * 8 + wave): s_memrealtime (100 MHz) at the start and at the end of its life
struct Stamps
	{
	unsigned long long last, acc[12], t_begin, r_begin;
	__device__ __forceinline__ static unsigned long long realtime()
		{
		unsigned long long t;
		asm volatile( "s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"( t ) :: "memory" );
		return t;
		}
	__device__ __forceinline__ static unsigned long long now()
		{
		unsigned long long t;
		__builtin_amdgcn_sched_barrier( 0 );
		asm volatile( "s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"( t ) :: "memory" );
		__builtin_amdgcn_sched_barrier( 0 );
		return t;
		}
#ifdef FLANHIP_STAMPS_CLOCK_ONLY
	// nothing is kept across the kernel (a live value would cost the frame loop registers): the start goes to memory at once
	__device__ __forceinline__ void init()
		{
		const unsigned long long r = realtime();
		const unsigned w = blockIdx.x * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
		if( ( threadIdx.x & 63 ) == 0 && w < 4096 ) g_stamp_span[2 * w] = r;
		}
#else
	__device__ __forceinline__ void init() { for( int i = 0; i < 12; ++i ) acc[i] = 0; r_begin = realtime(); last = now(); t_begin = last; }
#endif
#ifdef FLANHIP_STAMPS_CLOCK_ONLY
	// clock-only form: the frame loop is the product's (no stamp, no fence inside it); only the wavefront's life is measured in both clocks
	__device__ __forceinline__ void operator()( int ) const {}
#else
	__device__ __forceinline__ void operator()( int i ) { const unsigned long long t = now(); acc[i] += t - last; last = t; }
#endif
#ifdef FLANHIP_STAMPS_CLOCK_ONLY
	__device__ __forceinline__ void flush( int lane )
		{
		const unsigned long long r = realtime();
		const unsigned w = blockIdx.x * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
		if( lane == 0 && w < 4096 ) { g_stamp_span[2 * w + 1] = r; atomicAdd( &g_stamp_acc[15], 1ull ); }
		}
#else
	__device__ __forceinline__ void flush( int lane )
		{
		const unsigned long long t_end = now(), r_end = realtime();
		if( lane == 0 )
			{
			for( int i = 0; i < 12; ++i ) atomicAdd( &g_stamp_acc[i], acc[i] );
			atomicAdd( &g_stamp_acc[12], t_end - t_begin );                    // shader-clock ticks of the wavefront's life
			atomicAdd( &g_stamp_acc[13], r_end - r_begin );                    // the same span in 100 MHz ticks
			atomicAdd( &g_stamp_acc[15], 1ull );
			const unsigned w = blockIdx.x * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 );
			if( w < 4096 ) { g_stamp_span[2 * w] = r_begin; g_stamp_span[2 * w + 1] = r_end; }
			}
		}
#endif
	};
#else
struct Stamps
	{
	__device__ __forceinline__ void init() {}
	__device__ __forceinline__ void operator()( int ) const {}
	__device__ __forceinline__ void flush( int ) {}
	};
#endif

// (round_half_away_v, polar_tail / polar_v: pv_math.h)

// Before a chain publishes its head's tag from inside the frame loop (k_synthesize_v2 / _v3): every store this wavefront has issued has retired.
// FLANHIP_PUBLISH_DRAIN=0 is the A/B partner (round 5's form: the compiler's counted wait for the next row as the only proof)
#ifndef FLANHIP_PUBLISH_DRAIN
#define FLANHIP_PUBLISH_DRAIN 1
#endif
__device__ __forceinline__ void publish_drain()
	{
	if constexpr( FLANHIP_PUBLISH_DRAIN != 0 ) asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
	else asm volatile( "" ::: "memory" );
	}

// =================================================================================================================
// Audio::convert_to_PV (Conversions/AudioPV.cpp:12-78), dft 2048
// =================================================================================================================
// ABL (A/B builds only, 0 in the product): leave a phase out to see what it costs in place -- 1 the FFT passes, 2 atan2 / magnitude,
// 4 the wrap arithmetic, 8 the MF stores, 16 the sample loads, 32 the mirror read; 64: the MFs of two pairs in one 16-byte store and
// 128: two sample pairs in one 16-byte load; 1024: plain instead of non-temporal MF stores; 256: every frame of a chain stored over the chain's FIRST row (the same store
// instructions, no stream of fresh lines to HBM); 512: every frame loaded from the chain's first frame (TIMING ONLY: wrong places)
template<int WAVES, bool SUMS, int NV, int ABL = 0>
__global__ __launch_bounds__( 64 * WAVES ) void k_analyze_v2( AnalyzeParams p, FastTables tb )
	{
	using L = V2Lds;
	constexpr int C = 1024, E = 16, H = 8, NT = 64 * WAVES;
	constexpr int NP = NV / 2;                                                  // bin pairs evaluated together
	static_assert( NV == 2 || NV == 4 || NV == 8 || NV == 16, "1, 2, 4 or 8 pairs at a time" );
	typedef float VB __attribute__(( ext_vector_type( NV ) ));
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int W = p.window_size, hop = p.hop;
	const int cancel_word = cancel_peek( p.cancel );                           // travels under the table loads; a wavefront that finds it set walks no chain

	// ---- tables (block-wide) --------------------------------------------------------------------------------------------
	for( int i = tid; i < 240; i += NT ) s[L::TW1 + i] = tb.tw1[i];
	for( int i = tid; i < 768; i += NT ) s[L::TW3 + i] = tb.tw3[i];
	for( int i = tid; i < 512; i += NT ) { const cf w = tb.w2[i]; s[L::W2H + i] = mk( 0.5f * w.x, 0.5f * w.y ); }
		{
		float * win = reinterpret_cast<float*>( s + L::WIN );
		for( int i = tid; i < 2 * C; i += NT ) win[i] = ( i < W ) ? p.window[i] : 0.0f;          // AudioPV.cpp:60,65
		// bin frequency (PVBuffer.cpp:443-446: the division by dft, a power of two, is exactly a multiplication) and expected phase
		// advance (phase_vocoder.cpp:47) of the pair ( k, C - k )
		v4f_t * kc = reinterpret_cast<v4f_t*>( s + L::KC );
		const float rdft = 1.0f / float( 2 * C );
		for( int k = tid; k < 512; k += NT )
			{
			const float bk = float( k ) * p.sample_rate * rdft, bm = float( C - k ) * p.sample_rate * rdft;
			kc[k] = v4f_t{ bk, div_c( bk, p.ar_div ) * FLANHIP_PI2_F, bm, div_c( bm, p.ar_div ) * FLANHIP_PI2_F };
			}
		}
	__syncthreads();
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const cf * s_w2 = s + L::W2H + lane;
	const cf * s_win = s + L::WIN + lane;
	const v4f_t * s_kc = reinterpret_cast<const v4f_t*>( s + L::KC ) + lane;
	cf * buf = s + L::BUF + wave * L::BUF_LEN;
	// (Two wavefronts share a SIMD, one of wavefronts 0-3 and one of 4-7 of the block, and the older one wins: wavefronts 0-3 of every block
	// finish after ~112 us, 4-7 after ~129 us, profiles/r03_b_wave_spans.txt.  s_setprio 1 for the younger half -- for the whole chain, for its
	// first half, or whenever it is behind its partner's published frame count -- changed neither the split nor the launch's time: not kept.)
	// a block is a GROUP: WAVES consecutive chains of ONE channel (the last group of a channel may be short: its spare wavefronts idle).
	// No wavefront leaves early: the fused epilogue below meets at a block barrier.
	const int groups = ( p.chains_per_channel + WAVES - 1 ) / WAVES;
	const int channel = int( blockIdx.x ) / groups, group = int( blockIdx.x ) % groups;
	const int chain_in_channel = group * WAVES + wave;
	const bool active = chain_in_channel < p.chains_per_channel && !cancel_seen( cancel_word );
	const int64_t chain = int64_t( channel ) * p.chains_per_channel + ( active ? chain_in_channel : 0 );
	// frame numbers are ints here ( F <= n / hop + 1 and the host admits n < 2^31 - 8192 on this path)
	const int t0 = ( active ? chain_in_channel : 0 ) * p.L;
	const int t1 = int( min( int64_t( t0 ) + p.L, p.F ) );
	const float * x = p.audio + int64_t( channel ) * p.n;
	const int n32 = int( p.n );
	// Addresses: everything a block touches lies within ( WAVES L + 1 ) frames of the block's first halo frame tb0, so a global access is a
	// SCALAR base (channel and group come from blockIdx) plus a 32-bit per-lane byte offset -- one VGPR per stream instead of a 64-bit
	// pointer pair and 64-bit vector arithmetic per frame.  (The bases may point in front of the buffers for a channel's first group; no
	// access does: the clamped loads of edge frames keep absolute addresses.)
	const int tb0 = group * WAVES * p.L - 1;
	const char * const xb = reinterpret_cast<const char*>( x ) + ( int64_t( hop ) * tb0 - W / 2 ) * 4;
	char * const rb = reinterpret_cast<char*>( p.out + ( int64_t( channel ) * p.F + tb0 ) * ( C + 1 ) );
	const unsigned lane8 = 8u * unsigned( lane );
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                // phase_vocoder.cpp:37
	// (the run-time constants of the per-bin arithmetic -- analysis_rate and its reciprocal pair -- are wave-uniform and sit in SGPRs; forcing
	// them into VGPRs, because runs of VALU instructions with SGPR operands are slow ones in the issue microbenchmark, measured +-2 % in
	// two A/B runs of either sign: in the noise, not kept)
	const float k_ar = p.analysis_rate;
	const DivC k_ard = p.ar_div;
	const int padl = lane + ( lane >> 4 );
	const cf * mirror = buf + ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );   // mirror[-68 q] = slot PAD( C - lane - 64 q )
	struct __attribute__(( packed, aligned( 4 ) )) f2u { float x, y; };      // a sample pair at any 4-byte aligned address

	// state that crosses frames: previous phases (phase_vocoder.cpp:45) of the lane's 8 pairs and of bin C/2
	float prevk[H], prevm[H], prevx = 0.0f;
	#pragma unroll
	for( int q = 0; q < H; ++q ) { prevk[q] = 0.0f; prevm[q] = 0.0f; }        // AudioPV.cpp:44
	// fused round trip: per-chain sums of the phase increments convert_to_audio will integrate (phase_vocoder.cpp:57-58)
	double sumk[SUMS ? H : 1], summ[SUMS ? H : 1], sumx = 0.0;
	if constexpr( SUMS )
		{
		#pragma unroll
		for( int q = 0; q < H; ++q ) { sumk[q] = 0.0; summ[q] = 0.0; }
		}
	// running maximum of the magnitudes (Inf / NaN scan) taken on their BIT PATTERNS: a magnitude is never negative, so unsigned order is float
	// order with Inf and every NaN on top -- and an integer maximum needs no canonicalising v_max_f32 x, x per operand the way fmaxf does
	unsigned mmax = 0u;
	// Bin C/2 pairs with itself and would cost every lane ~70 instructions per frame for ONE bin.  Instead lane ( i & 63 ) keeps Z[C/2] of
	// the chain's i-th frame, and once per 64 frames (and at the chain's end) the wavefront works the batch off, one frame per lane:
	// the same arithmetic, previous phase from the lane below, the fp64 sum taken in frame order.
	cf ring = mk( 0.0f, 0.0f );

	// does every frame this chain touches (its halo frame included) lie inside the signal?  (AudioPV.cpp:52-62 needs no bounds then)
	const int tfirst = t0 > 0 ? t0 - 1 : t0;
	// Does frame t lie inside the signal with all of its 2 C samples (AudioPV.cpp:52-62 needs no bounds then)?  Decided PER FRAME: the first
	// and the last chain of a channel have two to four frames that reach outside, and a launch lasts as long as its slowest wavefront -- with
	// the clamped-load body chosen per chain those 16 chains ran 17 % longer than the 2032 others and set the launch's time
	// (profiles/r03_b_wave_spans.txt).
	const bool w_whole = ( W & 127 ) == 0;
	auto frame_inside = [&]( int t ) { return w_whole && hop * t - W / 2 >= 0 && hop * t - W / 2 + 2 * C <= n32; };   // ( hop t <= n: no overflow)

	Stamps st;
	st.init();
	cf z[E];
	auto run_chain = [&]()
		{
		// sample pair i = lane + 64 q of frame t (fast_tag: the frame lies inside the signal)
		auto load_pair = [&]( int t, int q, auto fast_tag ) -> cf
			{
			constexpr bool FAST = decltype( fast_tag )::value;
			const int start = hop * ( ( ABL & 512 ) ? tfirst : t ) - W / 2;
			if constexpr( ( ABL & 16 ) != 0 ) return mk( float( start ) * 1e-9f + 0.25f, float( q ) );
			else if constexpr( ( ABL & 128 ) != 0 && FAST )            // (interior chains only: the edge chains keep their clamped loads)
				{
				struct __attribute__(( packed, aligned( 4 ) )) f4u { float x, y, z, w; };
				const f4u v = *reinterpret_cast<const f4u*>( x + start + 4 * lane + 256 * ( q / 2 ) );
				return ( q & 1 ) ? mk( v.z, v.w ) : mk( v.x, v.y );
				}
			else if constexpr( FAST )
				{
				// (two offsets 4096 bytes apart: the instruction's immediate reaches 4095)
				const unsigned off = unsigned( hop * ( ( ( ABL & 512 ) ? tfirst : t ) - tb0 ) ) * 4u + lane8 + ( q >= 8 ? 4096u : 0u );
				const f2u v = *reinterpret_cast<const f2u*>( xb + off + 512 * ( q & 7 ) );
				return mk( v.x, v.y );
				}
			else
				{
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
			st( 0 );                                                              // 0: wait for the samples, window
			if constexpr( ( ABL & 1 ) == 0 )
			fft_fast<10>( z, buf, s_tw1, s_tw3, lane, st );                       // 1, 2: passes 0, 1 (each with the transpose after it)
			st( 3 );                                                              // 3: pass 2
			#pragma unroll
			for( int q = H; q < E; ++q ) buf[padl + 68 * q] = z[q];
			wave_sync();
			};

		// The loop is ROTATED: an iteration is [ per-bin work of frame t, which also requests frame t + 1's samples and stores frame t's
		// MFs ] then [ wait for those samples, window, transform of frame t + 1 ].  Request, stores and wait sit in one straight line
		// of code, so the wait is a counted one (s_waitcnt vmcnt(9 ..)) that leaves the stores in flight.  With the wait at the TOP of
		// a loop the compiler has to merge the state of the loop's entry edge (no stores behind the loads) with the back edge's and
		// drains the whole queue, stores included, once per frame (measured: a quarter of the kernel's time).
		// HALO: frame t0 - 1, of which only the phases are wanted (phase_vocoder.cpp:45 leaves them in phase_buffer): a compile-time
		// switch, not a branch.
		auto bins_of_frame = [&]( int t, int tn, int fi, auto halo_tag, auto next_fast )
			{
			constexpr bool halo = decltype( halo_tag )::value;
			const cf z512 = buf[544];                                             // Z[ C/2 ], slot PAD( 512 )
			const cf z0 = z[0];                                                   // lane 0: Z[0]
			#pragma unroll
			for( int q = H; q < E; ++q ) z[q] = load_pair( tn, q, next_fast );    // the upper half is in LDS now: its registers are free
			st( 4 );                                                              // 4: mirror exchange written, upper half of the next frame requested
			const unsigned roff = unsigned( ( ( ABL & 256 ) ? t0 : t ) - tb0 ) * unsigned( ( C + 1 ) * 8 );
			cf * row = reinterpret_cast<cf*>( rb + roff );
			cf * rowk = reinterpret_cast<cf*>( rb + ( roff + lane8 ) );
			cf * rowm = reinterpret_cast<cf*>( rb + ( roff + unsigned( C * 8 ) - lane8 ) );
			// the MFs of the frame leave together at its end, BEHIND every request for the next frame's samples: memory operations retire
			// in issue order, so a sample load issued after a store would not count as back before that store has been acknowledged
			// (microseconds while the chip streams writes), and the transform of the next frame would wait for it
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
					const cf zm = ( ABL & 32 ) ? zk : mirror[-68 * q];              // lane 0, q = 0 reads an unused slot: overridden below
					const cf w = s_w2[64 * q];                                      // 0.5 exp( -2 pi i k / 2C )
					const v4f_t kc = s_kc[64 * q];
					z[q] = load_pair( tn, q, next_fast );                           // Z[k] is consumed: next frame's samples take its place
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
				if constexpr( ( ABL & 2 ) == 0 ) polar_v( re, im, phase, m ); else { phase = re; m = im; }
				#pragma unroll
				for( int i = 0; i < NP; ++i ) { prevk[g * NP + i] = phase[i]; prevm[g * NP + i] = phase[NP + i]; }   // :45
				if constexpr( !halo )
					{
					VB f;
					if constexpr( ( ABL & 4 ) == 0 )
						{
						const VB phase_diff = phase - pv;                            // == float( double(phase) - double(prev) ), :44
						const VB delta_phase = phase_diff - expd;                    // :47-48
						VB wrapped = delta_phase;
						if( use_wrapping ) wrapped = delta_phase - vsplat<VB>( FLANHIP_PI2_F ) * round_half_away_v( div_pi2_v( delta_phase ) );   // :39-42,49
						VB war;                                                       // (element by element: a splat of a run-time constant costs one SGPR per element)
						#pragma unroll
						for( int i = 0; i < NV; ++i ) war[i] = wrapped[i] * k_ar;
						f = binf + div_pi2_v( war );                                  // :50-52
						}
					else f = phase - pv + binf * expd;
					#pragma unroll
					for( int i = 0; i < NP; ++i )
						{
						const int q = g * NP + i;
						outk[q] = cf{ m[i], f[i] };
						outm[q] = cf{ m[NP + i], f[NP + i] };
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
				st( 5 + ( g & 3 ) );                                              // 5..8: the groups of bins (modulo 4)
				}
			if constexpr( !halo )
				{
				#pragma unroll
				for( int q = 0; q < H; ++q )
					{
					if constexpr( ( ABL & 64 ) != 0 )
						{
						if( ( q & 1 ) == 0 )
							{
							*reinterpret_cast<v4f_t*>( reinterpret_cast<float*>( row ) + 4 * ( lane + 64 * ( q / 2 ) ) ) = v4f_t{ outk[q].x, outk[q].y, outk[q + 1].x, outk[q + 1].y };
							*reinterpret_cast<v4f_t*>( reinterpret_cast<float*>( row ) + 4 * ( lane + 64 * ( q / 2 + 4 ) ) ) = v4f_t{ outm[q].x, outm[q].y, outm[q + 1].x, outm[q + 1].y };
							}
						}
					else if constexpr( ( ABL & 1024 ) != 0 )
						{
						rowk[64 * q] = outk[q];
						rowm[-64 * q] = outm[q];
						}
					else if constexpr( ( ABL & 8 ) == 0 )
						{
						// non-temporal: the PV is written once and read by another kernel much later; measured, these stores are acknowledged
						// sooner than plain ones while the chip streams writes (the kernel 9 % faster, its memory traffic alone 27 %)
						__builtin_nontemporal_store( outk[q], rowk + 64 * q );
						__builtin_nontemporal_store( outm[q], rowm - 64 * q );
						}
					else asm volatile( "" :: "v"( outk[q].x ), "v"( outk[q].y ), "v"( outm[q].x ), "v"( outm[q].y ) );
					}
				}
			ring = ( lane == ( fi & 63 ) ) ? z512 : ring;                          // Z[ C/2 ] of the chain's fi-th frame waits in lane fi % 64
			wave_sync();
			st( 9 );                                                              // 9: bin C/2
			};

		// the batch of bin C/2: lane j holds frame tb + j, j < nb (frame tfirst, the halo, only lends its phase)
		auto flush_half_bin = [&]( int tb, int nb )
			{
			const int t = tb + lane;
			const bool valid = lane < nb && t >= t0;
			const float re = ring.x, im = -ring.y;                                // X = conj Z[ C/2 ]
			const float phase = atan2_fast( im, re );
			float pvx = __shfl_up( phase, 1 );                                    // the frame before: the lane below ...
			pvx = ( lane == 0 ) ? prevx : pvx;                                    // ... or the last frame of the batch before (zero at a channel's start, AudioPV.cpp:44)
			prevx = __shfl( phase, nb - 1 );
			const float bx = float( C / 2 ) * p.sample_rate * ( 1.0f / float( 2 * C ) );
			const float phase_diff = phase - pvx;
			const float delta_phase = phase_diff - div_c( bx, p.ar_div ) * FLANHIP_PI2_F;
			const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * round_half_away( div_pi2( delta_phase ) ) : delta_phase;
			const float f = bx + div_pi2( wrapped * p.analysis_rate );
			const float m = magnitude_scaled( re, im );
			if constexpr( ( ABL & 8 ) == 0 )
				{
				if( valid ) __builtin_nontemporal_store( mk( m, f ), reinterpret_cast<cf*>( rb + ( unsigned( ( ( ABL & 256 ) ? t0 : t ) - tb0 ) * unsigned( ( C + 1 ) * 8 ) + unsigned( C / 2 * 8 ) ) ) );
				}
			else asm volatile( "" :: "v"( m ), "v"( f ) );
			if constexpr( SUMS )
				{
				const float term = div_c( f, p.ar_div ) * FLANHIP_PI2_F;             // phase_vocoder.cpp:57-58
				const int j0 = __builtin_amdgcn_readfirstlane( ( tb < t0 ) ? 1 : 0 ), j1 = __builtin_amdgcn_readfirstlane( nb );   // (uniform: scalar loop, v_readlane)
				for( int j = j0; j < j1; ++j ) sumx += double( __uint_as_float( __builtin_amdgcn_readlane( __float_as_uint( term ), j ) ) );   // in frame order, like every other bin's sum
				mmax = valid ? max( mmax, __float_as_uint( m ) ) : mmax;
				}
			};

		constexpr std::true_type inside{};
		constexpr std::false_type outside{};
		if( frame_inside( tfirst ) )
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) z[q] = load_pair( tfirst, q, inside );
			transform_frame( tfirst, inside );
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < E; ++q ) z[q] = load_pair( tfirst, q, outside );
			transform_frame( tfirst, outside );
			}
		int fi = 0;
		if( t0 > 0 )
			{
			if( frame_inside( t0 ) ) { bins_of_frame( t0 - 1, t0, fi, std::true_type{}, inside ); transform_frame( t0, inside ); }
			else { bins_of_frame( t0 - 1, t0, fi, std::true_type{}, outside ); transform_frame( t0, outside ); }
			++fi;
			}
		for( int t = t0; t < t1; ++t )
			{
			const int tn = min( t + 1, t1 - 1 );                               // (the last frame requests itself again: nobody waits for it)
			const bool more = t + 1 < t1;
			auto rest_of_step = [&]( auto next_fast )
				{
				bins_of_frame( t, tn, fi, std::false_type{}, next_fast );
				++fi;
				if( ( fi & 63 ) == 0 || !more ) flush_half_bin( t + 1 - ( ( ( fi - 1 ) & 63 ) + 1 ), ( ( fi - 1 ) & 63 ) + 1 );
				if( more ) transform_frame( t + 1, next_fast );
				};
			if( frame_inside( tn ) ) rest_of_step( inside ); else rest_of_step( outside );
			}
		};
	if( active ) run_chain();
	st.flush( lane );

	if constexpr( SUMS )
		{
		// the chain's sums, folded like phase_vocoder.cpp:59, go to the workspace (what k_phase_sums2 would compute) and -- staged in this
		// wavefront's now idle transform buffer -- into the group's total: with one total per group of 8 chains the synthesis kernel can
		// work out its own carries (a few dozen additions per bin) and the scan kernel between the two is not launched at all
		double * stage = reinterpret_cast<double*>( buf );                        // 1025 doubles = 8200 B of the buffer's 8712
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
			// no clearing pass: the flag word is "set" when it equals this launch's epoch (written beside it by chain 0)
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
					const double v = run + reinterpret_cast<const double*>( s + L::BUF + w * L::BUF_LEN )[bin];
					run = ( __builtin_fabs( v ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( v ) : fold_phase_any( v );
					}
				gdst[bin] = run;
				}
			}
		}
	}

// =================================================================================================================
// PV::convert_to_audio (Conversions/AudioPV.cpp:86-139), dft 2048; HOPQ = hop / 128 for the hops 128 / 256 / 512 / 1024
// (overlap-add accumulator in registers).  Same ideas as k_analyze_v2:
//   * a lane owns the bin pairs ( k, C - k ), k = lane + 64 q, q < 8: the merge of X[k], X[C-k] into the half-size spectrum gives Z[k] and
//     Z[C-k] from shared sums and one twiddle product; only Z[C-k] crosses LDS (to the lane that owns it in the transform's layout);
//   * the frame loop is rotated -- [ request row t + 1 | transform, window, overlap-add, stores of frame t | wait, per-bin work of
//     row t + 1 ] -- so that the wait for the MF row is a counted one that leaves the output stores in flight;
//   * the MF rows are read once: non-temporal loads.
// =================================================================================================================
// 8-byte loads / stores that other XCDs' wavefronts see inside a launch (agent scope: the L2s of two XCDs are not coherent for ordinary accesses)
__device__ __forceinline__ void st_agent( cf * p, cf v )
	{
	unsigned long long bits; __builtin_memcpy( &bits, &v, 8 );
	__hip_atomic_store( reinterpret_cast<unsigned long long*>( p ), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
	}
__device__ __forceinline__ cf ld_agent( const cf * p )
	{
	const unsigned long long bits = __hip_atomic_load( reinterpret_cast<const unsigned long long*>( p ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
	cf v; __builtin_memcpy( &v, &bits, 8 );
	return v;
	}

struct V2LdsSyn
	{
	static constexpr int C = 1024;
	static constexpr int TW1 = 0;
	static constexpr int TW3 = TW1 + 240;
	static constexpr int W2S = TW3 + 768;                  // [512]   ( cos, sin )( pi k / C ) = exp( +2 pi i k / 2C ), k < C/2
	static constexpr int WIN = W2S + 512;                  // [2048 floats] hann * window_scale
	static constexpr int BUF = WIN + 1024;
	static constexpr int BUF_LEN = C + C / 16 + 1;
	static constexpr size_t bytes( int waves ) { return size_t( BUF + waves * BUF_LEN ) * 8; }
	};

template<int WAVES, int HOPQ, int ABL = 0>
__global__ __launch_bounds__( 64 * WAVES ) void k_synthesize_v2( SynthParams p, FastTables tb )
	{
	using L = V2LdsSyn;
	constexpr int C = 1024, E = 16, H = 8, NT = 64 * WAVES;
	static_assert( HOPQ == 0 || HOPQ == 1 || HOPQ == 2 || HOPQ == 4 || HOPQ == 8, "hop 128 / 256 / 512 / 1024, or 0: any hop <= window, any window <= dft" );
	// RING (HOPQ = 0, round 5): the overlap-add accumulator as a ring of `window` floats per wavefront in LDS instead of registers -- hops that are no multiple
	// of 128 (300, 441 ...), windows that are none (2000, 1800 ...): those ran the round-1 kernel (k_synthesize_fast) until now.  Samples leave one by
	// one (4-byte stores), the chains' overlaps through k_ola_fixup.
	constexpr bool RING = HOPQ == 0;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cf * s = reinterpret_cast<cf*>( smem );
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int W = p.window_size;
	const int cancel_word = cancel_peek( p.cancel );
	for( int i = tid; i < 240; i += NT ) s[L::TW1 + i] = tb.tw1[i];
	for( int i = tid; i < 768; i += NT ) s[L::TW3 + i] = tb.tw3[i];
	for( int i = tid; i < 512; i += NT ) { const cf w = tb.w2[i]; s[L::W2S + i] = mk( w.x, -w.y ); }
		{
		float * win = reinterpret_cast<float*>( s + L::WIN );
		for( int i = tid; i < 2 * C; i += NT ) win[i] = ( i < W ) ? p.window[i] * p.window_scale : 0.0f;      // AudioPV.cpp:102
		}
	if( !p.group_sums ) __syncthreads();                                        // (with group sums the barrier of the carry prologue below serves the tables too)
	const cf * s_tw1 = s + L::TW1;
	const cf * s_tw3 = s + L::TW3;
	const cf * s_w2 = s + L::W2S + lane;
	const cf * s_win = s + L::WIN + lane;
	cf * buf = s + L::BUF + wave * L::BUF_LEN;

	// a block is a GROUP of WAVES consecutive chains of one channel (see k_analyze_v2); spare wavefronts of a channel's last group idle
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
	float * ring = reinterpret_cast<float*>( s + L::BUF + WAVES * L::BUF_LEN ) + wave * wpad;      // RING only
	if constexpr( RING ) { for( int i = lane; i < wpad; i += 64 ) ring[i] = 0.0f; }
	int ring_base = 0;
	float * out1 = p.out + int64_t( channel ) * p.out_len;
	cf * out2 = reinterpret_cast<cf*>( out1 );
	cf * head2 = reinterpret_cast<cf*>( p.head + chain * p.head_len );
	const int64_t chain_start = int64_t( hop ) * t0 - W / 2;
	const int64_t own_start = chain_in_channel == 0 ? INT64_MIN : chain_start + p.head_len;
	// Addresses of the frame loop: a SCALAR base per block (channel and group come from blockIdx) plus a 32-bit per-lane byte offset -- the
	// MF rows and the finished output a block touches lie within WAVES L frames of its first frame tb0 (see k_analyze_v2)
	const int64_t tb0 = int64_t( group ) * WAVES * p.L;
	const int relf0 = ( active ? wave : 0 ) * p.L;                              // this chain's first frame, counted from tb0
	const int nf = int( t1 - t0 );
	const char * const pvb = reinterpret_cast<const char*>( p.pv + ( int64_t( channel ) * p.F + tb0 ) * ( C + 1 ) );
	char * const ob = reinterpret_cast<char*>( out1 ) + ( int64_t( hop ) * tb0 - W / 2 ) * 4;   // (in front of the buffer for a channel's first group: its chain 0 takes the general emit_step)
	const unsigned lane8 = 8u * unsigned( lane );
	// every step a chain emits once its head is written lies inside the output, except for a channel's chain 0 (positions below 0)
	const bool plain_emit = __builtin_amdgcn_readfirstlane( int( chain_in_channel != 0 ) ) != 0 && ( ABL & 4 ) == 0;
	const int padl = lane + ( lane >> 4 );
	cf * mirror = buf + ( C * 17 ) / 16 - ( lane + ( ( lane + 15 ) >> 4 ) );      // mirror[-68 q] = slot PAD( C - lane - 64 q )

	const DivC k_ard = p.ar_div;
	cf acc[E];                                                                  // overlap-add accumulator: acc[q] <-> samples pos + 128 q + 2 lane (+1)
	#pragma unroll
	for( int q = 0; q < E; ++q ) acc[q] = mk( 0.0f, 0.0f );

	// one 128-sample step of finished (or partial) output leaves the chain (positions are even: hop and W/2 are multiples of 64).
	// Exactly one store instruction per step, never inside a branch: lanes that fall outside the output are pointed at a 512-byte dump
	// area in the workspace, so the number of stores in flight behind the row request is static (counted wait, see the loop)
	cf * dump2 = reinterpret_cast<cf*>( p.dump ) + lane;
	const bool fix = !RING && p.fix_state != nullptr;                           // this launch adds the chains' overlaps itself (below)
	float * head1 = p.head + chain * p.head_len;
	int64_t pos_of_ring = chain_start;                                           // RING: ring[ring_base] <-> this absolute sample
	// RING: the oldest `count` samples of the ring leave (and are cleared); sample by sample like the generic kernels (pv_kernels.h)
	auto ring_emit = [&]( int count )
		{
		if( count <= W && pos_of_ring >= own_start && pos_of_ring >= 0 && pos_of_ring + count <= p.out_len )
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
				v = ring[j]; ring[j] = 0.0f;
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
		if( ( ABL & 4 ) || ( a0 >= own_start && !( a >= 0 && a < p.out_len ) ) ) dst = dump2;           // ABL 4 (timing only): every store to the dump area
		if( fix && a0 < own_start ) st_agent( dst, v );                          // (the head another wavefront may come to add up: see the end of the kernel)
		else *dst = v;
		};
	// MF row of frame t: ( m, f ) of the lane's pairs and of bin C/2
	cf mfk[H], mfm[H], mfx;
	auto load_row = [&]( int fr )                                               // fr: the frame, counted from tb0
		{
		const unsigned ro = unsigned( ( ABL & 2 ) ? relf0 : fr ) * unsigned( ( C + 1 ) * 8 );   // ABL 2 (timing only): a hot row
		const cf * row = reinterpret_cast<const cf*>( pvb + ro );
		const cf * rowk = reinterpret_cast<const cf*>( pvb + ( ro + lane8 ) );
		const cf * rowm = reinterpret_cast<const cf*>( pvb + ( ro + unsigned( C * 8 ) - lane8 ) );
		#pragma unroll
		for( int q = 0; q < H; ++q )
			{
			mfk[q] = ( ABL & 1 ) ? rowk[64 * q] : __builtin_nontemporal_load( rowk + 64 * q );
			mfm[q] = ( ABL & 1 ) ? rowm[-64 * q] : __builtin_nontemporal_load( rowm - 64 * q );
			}
		mfx = ( ABL & 1 ) ? row[C / 2] : __builtin_nontemporal_load( row + C / 2 );
		};

	// phase_buffer (AudioPV.cpp:105) on entry to the chain, of the lane's pairs and of bin C/2
	double phk[H], phm[H], phx;
	if( p.group_sums )
		{
		// no scan over the chains ran: `carry` still holds the chains' own sums, group_carry the running phase on entry to every group of 8 chains
		// (a scan over the producer's group totals: an eighth of the elements).  The running phase on entry to this chain = that, then the chains of
		// this group before this one, added and folded in order (phase_vocoder.cpp:57-59 modulo pi2: the prefix k_phase_scan2 forms, associated group-wise).
		// x + y folded like phase_vocoder.cpp:59: the branch-free fold of the frame loop (pv_math.h) wherever it is exact, i.e. always but
		// for sums beyond 3e9 rad or NaN, which take the general routine
		auto fold = []( double r )
			{
			return ( __builtin_fabs( r ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( r ) : fold_phase_any( r );
			};
		// One thread per bin, its two or three bins side by side (independent dependency chains), along the chains of this group, leaving every
		// wavefront's carries in that wavefront's own transform buffer (1025 doubles of its 8712 bytes).  Every load goes out ahead of the
		// dependent additions: one memory round trip.
		const double * gs = ( p.group_carry ? p.group_carry : p.group_sums ) + int64_t( channel ) * groups * ( C + 1 );
		const double * sums0 = p.carry + ( int64_t( channel ) * p.chains_per_channel + int64_t( group ) * WAVES ) * ( C + 1 );   // the first chain of this group
		const int live = min( WAVES, p.chains_per_channel - group * WAVES );
		constexpr int NB = ( C + NT ) / NT;                                       // bins per thread: 3 for 512 threads (the third only for thread 0)
		int bins_of[NB]; bool has[NB]; double run[NB];
		#pragma unroll
		for( int b = 0; b < NB; ++b ) { bins_of[b] = tid + NT * b; has[b] = bins_of[b] <= C; if( !has[b] ) bins_of[b] = C; run[b] = 0.0; }
		double vc[NB][WAVES];                                                     // the chains of this group: requested first, used last
		#pragma unroll
		for( int b = 0; b < NB; ++b )
			{
			#pragma unroll
			for( int w = 0; w < WAVES; ++w ) vc[b][w] = ( w < live ) ? sums0[int64_t( w ) * ( C + 1 ) + bins_of[b]] : 0.0;
			}
		if( p.group_carry )
			{
			#pragma unroll
			for( int b = 0; b < NB; ++b ) run[b] = gs[int64_t( group ) * ( C + 1 ) + bins_of[b]];   // the running phase on entry to this group (k_phase_scan2<SEG, true>)
			}
		if( active ) load_row( relf0 );                                           // the first MF row travels while the carries are worked out
		if( !p.group_carry )
			{
			// few groups per channel (the host's choice): no scan over the group totals was launched -- this group adds up the totals of the groups
			// before it itself, 16 loads per bin in flight (group g reads g totals: O(groups^2) bytes in all, cheaper than a kernel up to ~40 groups)
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
					for( int b = 0; b < NB; ++b ) run[b] = fold( run[b] + v[b][u] );      // + 0.0 past the end: fold( x ) of a folded x is x
					}
				}
			}
		#pragma unroll
		for( int w = 0; w < WAVES; ++w )
			{
			#pragma unroll
			for( int b = 0; b < NB; ++b )
				{
				if( has[b] ) reinterpret_cast<double*>( s + L::BUF + w * L::BUF_LEN )[bins_of[b]] = run[b];   // phase_buffer on entry to chain w of the group
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
		wave_sync();                                                              // own buffer only: read before the transform writes it
		}
	else if( active )
		{
		const double * carry = p.carry + chain * ( C + 1 );
		#pragma unroll
		for( int q = 0; q < H; ++q ) { phk[q] = carry[lane + 64 * q]; phm[q] = carry[C - lane - 64 * q]; }
		phx = carry[C / 2];
		}
	if( !active ) return;
	Stamps st;                                                                  // (diagnostic builds: the wavefront's life, tools/wave_spans.py)
	st.init();
	cf z[E];
	// inverse phase vocoder of the row in mfk / mfm / mfx (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61), merge of X[0..C] into the
	// half-size spectrum conj( A + i B ) (the forward transform of it is the conjugate of the inverse one): leaves z[] complete
	auto bins_of_row = [&]()
		{
		bool slow = false;
		// f / analysis_rate of the row's 17 bins under ONE test of the divisor's plan (div_c per bin is a scalar branch per bin: 17 islands of
		// three dependent instructions each instead of 17 interleaved chains)
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
			// a phase outside the range the fast helpers are exact for (or a NaN): the general routines for this frame
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
				// two pairs = four bins per iteration as one vector stream (pv_math.h)
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
				__builtin_amdgcn_sched_barrier( 0 );                            // four bins at a time: keeps the temporaries of 16 bins from overlapping
				}
			phx = fold_phase_loop( phx );
			float sn, cs;
			sincos_fast( float( phx ), sn, cs );
			xx = mk( mfx.x * cs, mfx.x * sn );
			}
		// ---- merge: Z[k] = conj( A + i B ), Z[C-k] likewise with A -> conj A, B -> -conj B ... written out per component below
		#pragma unroll
		for( int q = 0; q < H; ++q )
			{
			cf a = xk[q], b = xm[q];                                            // X[k], X[C-k]
			if( q == 0 ) { a.y = ( lane == 0 ) ? 0.0f : a.y; b.y = ( lane == 0 ) ? 0.0f : b.y; }   // c2r ignores Im X[0], Im X[C]
			const cf w = s_w2[64 * q];                                          // exp( +2 pi i k / N ): ( c, sgn )
			const float ax = a.x + b.x, ay = a.y - b.y;                         // A = X[k] + conj X[C-k]
			const float dx = a.x - b.x, dy = a.y + b.y;                         // D = X[k] - conj X[C-k]
			const float bx = __builtin_fmaf( w.x, dx, -( w.y * dy ) ), by = __builtin_fmaf( w.x, dy, w.y * dx );
			z[q] = mk( ax - by, -( ay + bx ) );                                 // Z[k]
			mirror[-68 * q] = mk( ax + by, ay - bx );                           // Z[C-k] (lane 0, q = 0: an unused slot)
			}
		if( lane == 0 ) buf[544] = mk( 2.0f * xx.x, 2.0f * xx.y );              // Z[C/2] = 2 X[C/2] (slot PAD( 512 ))
		wave_sync();
		#pragma unroll
		for( int q = H; q < E; ++q ) z[q] = buf[padl + 68 * q];
		wave_sync();
		};

	if( !p.group_sums ) load_row( relf0 );
	bins_of_row();
	int64_t pos = chain_start;
	int rel = 0;                                                                // pos - chain_start: the same number in every wavefront, a scalar
	// ---- the overlaps of neighbouring chains added here instead of by a launch of their own (k_ola_fixup: a launch and two round trips to memory
	// behind every convert_to_audio; p.fix_state set).  The W - hop samples at a boundary get the LAST partial sums of the chain before it (in its
	// `acc` when it ends) and the FIRST ones of the chain after it (in its `head` buffer since its first frames).  One word per boundary, tagged with
	// the launch's epoch, written by atomic exchange; whoever finds the other side's tag there adds the two halves, so no wavefront ever waits for
	// another and the order in which blocks are scheduled cannot matter.  Round 5: the head's owner publishes INSIDE its frame loop, as soon as the
	// loop's own counted wait has proven the head's stores acknowledged (memory operations retire in order) -- in a launch of one round that is
	// ~100 us before its neighbour ends; the tail's owner reads the word one frame before its last (the answer arrives under that frame's row wait),
	// requests the head under its LAST transform and adds it to its accumulator as it leaves: no exchange, no round trip at the end of the launch
	// (round 4's form -- both sides at their ends -- cost the launch the 5-7 us the separate kernel took).  The halves cross XCDs inside a launch:
	// written and read at agent scope (st_agent / ld_agent).  One addition per sample, tail + head, as k_ola_fixup does it: the same bits.
	const int tag_tail = p.fix_tag | 1, tag_head = p.fix_tag | 2;
	const int nsteps = p.head_len / 128;                                        // steps of 128 samples a boundary holds (W - hop, a multiple of 128 here)
	const bool has_head = chain_in_channel != 0, has_tail = !last_chain;
	int * const word_h = p.fix_state + chain, * const word_t = p.fix_state + ( chain + 1 );      // (used under `fix` only)
	const cf * const head_next = reinterpret_cast<const cf*>( p.head + ( chain + 1 ) * p.head_len ) + lane;
	const int i_pub = ( p.head_len + hop - 1 ) / hop;                           // the frame whose row wait proves the head's stores have landed
	int old_h = 0, seen_t = 0;                                                  // (lane 0's: what the head word held before this chain's tag; what the tail word holds)
	bool published = false;
	cf hx[E];                                                                   // the next chain's head (the last frame only: in the registers the MF rows leave)
	auto frame_step = [&]( int i, auto last_tag ) -> bool
		{
		constexpr bool LAST = decltype( last_tag )::value;
		bool have_head = false;
		if constexpr( LAST )
			{
			if( fix && has_tail )
				{
				have_head = __builtin_amdgcn_readfirstlane( seen_t ) == tag_head;        // the neighbour's head was complete a frame ago
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
		if constexpr( ( ABL & 8 ) == 0 )                                        // ABL 8 (timing only): no transform
		fft_fast<10>( z, buf, s_tw1, s_tw3, lane );
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
			ring_emit( hop );
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
		// HOPQ stores in either arm (the wait for the next row counts them).  The plain arm: past the chain's head every step goes to the
		// output proper, all of it inside -- a scalar base, one 32-bit offset, immediates; the general arm (head steps, chain 0) costs ~17
		// 64-bit vector instructions per step
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
				// the head's stores went out iterations ago and the wait for row i + 1 has retired them with everything else issued before that row was
				// requested -- but that is the compiler's counted wait, not a statement of this source: the queue is drained explicitly (once per chain:
				// this iteration's HOPQ stores are all that is in flight; ADVICE r05, interleaved A/B in profiles/r06_ab_publish_drain.txt)
				publish_drain();
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
		st.flush( lane );
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
			// a chain too short to have published from its loop: now, behind a drained queue
			asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
			if( lane == 0 ) old_h = __hip_atomic_exchange( word_h, tag_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
			}
		if( has_tail )
			{
			// this chain's tail meets the next chain's head
			cf * tail_next = reinterpret_cast<cf*>( p.tail + ( chain + 1 ) * p.head_len ) + lane;
			bool add = have_head;
			if( !add )
				{
				// the neighbour had not published a frame ago (a launch of several rounds, a chain of one frame): leave the tail where it will
				// find it, BEHIND a drained queue, and say so; if its tag has appeared meanwhile the addition is ours after all
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
			// this chain's head meets the previous chain's tail, which was there when the head's tag went out
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
	st.flush( lane );
	}

} // namespace flanhip
