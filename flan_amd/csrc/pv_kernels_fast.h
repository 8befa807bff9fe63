// pv_kernels_fast.h -- the register transform (fft_fast) and the pre-pass kernels shared by the tuned conversion kernels.  (Round 1's own kernels for
// dft 2048 / 4096, k_analyze_fast / k_synthesize_fast, lived here until round 6: every shape they served has a successor -- pv_kernels_v2.h, pv_kernels_eo.h --
// and the generic kernels of pv_kernels.h are the A/B predecessor of every tuned size.)  What follows describes the design they introduced:
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

// The same scan where a channel has FEW chains and a row has many bins (dft 8192 / 16384 team kernels: 64 / 32 chains per channel at 8 channels, 4097 / 8193
// bins): one thread per ( channel, bin ) holds the whole column -- every chain sum requested before the first dependent addition, no LDS, no block barrier,
// one trip to memory each way.  k_phase_scan2<16> took 25 us for the 64 x 4097 x 8 sums of (8192, 2048, 8192) (segments of 4 chains through LDS: two dependent
// memory round trips per thread); this takes the bandwidth's time.  Plain scans only (no carry_in / total_out: frame-range sharding keeps k_phase_scan2).
template<int MAXC>
__global__ __launch_bounds__( 256 ) void k_phase_scan_flat( SynthParams p )
	{
	const int k = blockIdx.x * 256 + threadIdx.x, channel = blockIdx.y;
	if( threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 )
		{
		if( p.nan_in && p.nan_flag && p.nan_in[0] == p.nan_in[2] && p.nan_in[2] != 0 ) atomicOr( p.nan_flag, 1 );
		if( p.expect_epoch && p.nan_in && p.nan_flag && p.nan_in[2] != p.expect_epoch ) atomicOr( p.nan_flag, 2 );   // the sums in this workspace are not the noted producer's
		if( p.skip_words ) const_cast<int*>( p.skip_words )[4] = 0;                // the sums become carries below: a handed-over pre-pass is good for one convert_to_audio
		}
	if( k >= p.num_bins ) return;
	const int n = p.chains_per_channel;
	double * c = p.carry + int64_t( channel ) * n * p.num_bins + k;
	double held[MAXC];
	#pragma unroll
	for( int i = 0; i < MAXC; ++i ) held[i] = ( i < n ) ? c[int64_t( i ) * p.num_bins] : 0.0;
	double run = 0.0;
	#pragma unroll
	for( int i = 0; i < MAXC; ++i )
		{
		if( i < n ) c[int64_t( i ) * p.num_bins] = run;                            // phase_buffer on entry to chain i
		const double v = run + held[i];
		run = ( __builtin_fabs( v ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_loop( v ) : fold_phase_any( v );
		}
	}

} // namespace flanhip
