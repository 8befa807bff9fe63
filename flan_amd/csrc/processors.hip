// processors.hip -- PV frame processors behind the C ABI: modify_time_base / stretch, modify_frequency_base / repitch,
// shape (reference: PV/PVModify.cpp:196-257, :273-385; PV/PV.cpp:421-458).
//
// These algorithms are order dependent by definition (a running, magnitude-weighted average per (channel, bin) over
// frames; a "louder wins, magnitudes add" rule per (channel, frame) over bins), so each kernel keeps the reference's
// sequential order along the dependent axis and spreads the independent axes over lanes:
//   k_modify_time      : lane = bin      (adjacent lanes -> adjacent bins: coalesced 8-byte MF traffic), loop over frames
//   k_modify_frequency : lane = frame    (rows are walked by one lane; the row stays in L2)
//   k_shape            : lane = element (no alignment) or wavefront = row (shift alignment: the sequential conflict rule is
//                        resolved through LDS keys, processors_common.h)
#include "processors_common.h"
#include "const_sum.h"
#include "pv_math.h"
#include <algorithm>

namespace flanhip {

// modify_time_base, PVModify.cpp:319-359 (linear Interpolator, Utility/Interpolator.cpp:50-56).
//
// The reference walks each (channel, bin) column frame pair by frame pair and ACCUMULATES into the output frames
// [ceil(l), ceil(r)) the pair maps to -- order dependent in general.  But where the map of a bin never runs backwards
// (r >= l for every pair: every stretch / slow-down / speed-up) those intervals are disjoint and tile
// [ceil(l_first), ceil(r_last)) without gaps: every output frame receives at most one contribution, onto the cleared
// value { 0, 0 } (:317).  Such columns are cut into `segments` runs of frame pairs handled by different threads, each thread
// OWNS the output frames of its pairs: it writes them once (the accumulation onto { 0, 0 } written out literally), zeros where
// the reference leaves a pair early (:350-351) and, at the two ends of the column, the frames no pair reaches -- no separate
// clearing pass and no read-modify-write.  Columns whose map does run backwards (flag set by the first blocks of k_modify_time_chains) are cleared
// and then walked by one thread in the reference order.
__global__ __launch_bounds__( 256 ) void k_modify_time( const MFd * in, int num_channels, int64_t F, int bins, float sr, float hop,
	const float * mod, int64_t Fo, MFd * out, const int * nonmono, int flag_tag, int segments, int64_t seg_len, int only_if_any, int interp, int * words, int epoch )
	{
	if( only_if_any && nonmono[bins] != flag_tag ) return;                          // k_modify_time_chains has done this PV (a flag is set when it holds this call's tag)
	// the chain sums k_modify_time_chains left are NOT this PV's, and neither is the NaN / Inf word its chain blocks may have set from frames this
	// kernel overwrites: both are taken back ([4] = 0 makes convert_to_audio's pre-pass scan the real PV again)
	if( words && blockIdx.x == 0 && threadIdx.x == 0 ) { words[0] = 0; words[2] = epoch; words[4] = 0; }
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	const int64_t columns = int64_t( num_channels ) * bins;
	if( idx >= columns * segments ) return;
	const int seg = int( idx / columns );
	const int channel = int( ( idx % columns ) / bins ), bin = int( idx % bins );
	int64_t f0 = 1 + seg * seg_len, f1 = min( F, f0 + seg_len );
	const bool sequential = nonmono[bin] == flag_tag;
	if( sequential ) { if( seg != 0 ) return; f0 = 1; f1 = F; }                    // order matters for this column: one thread, all pairs
	if( f0 >= f1 ) return;
	const MFd * ip = in + int64_t( channel ) * F * bins + bin;
	MFd * op = out + int64_t( channel ) * Fo * bins + bin;
	const float * mp = mod + bin;
	const int Fo32 = int( Fo );
	auto clear = [&]( int x0, int x1 ) { for( int x = x0; x < x1; ++x ) op[int64_t( x ) * bins] = MFd{ 0.0f, 0.0f }; };   // :317 clear_buffer
	MFd mf_l = ip[( f0 - 1 ) * bins];
	float at_l = time_to_frame( mp[( f0 - 1 ) * bins], sr, hop );
	if( sequential ) clear( 0, Fo32 );
	else if( f0 == 1 ) clear( 0, min( max( int( ceilf( at_l ) ), 0 ), Fo32 ) );   // before the first pair's interval
	for( int64_t frame = f0; frame < f1; ++frame )                                  // :328
		{
		const MFd mf_r = ip[frame * bins];
		const float at_r = time_to_frame( mp[frame * bins], sr, hop );            // :331
		const bool forward = at_r > at_l;                                       // :332
		const int start_frame = int( forward ? ceilf( at_l ) : floorf( at_l ) ); // :334
		const int end_frame   = int( forward ? ceilf( at_r ) : floorf( at_r ) ); // :335
		if( !sequential )
			{
			// forward or empty; frames outside [0, Fo) are skipped by :342 before anything is computed, so clamping is exact
			const int x1 = min( max( end_frame, 0 ), Fo32 );
			int x = min( max( start_frame, 0 ), Fo32 );
			for( ; x < x1; ++x )                                                    // :340
				{
				const float mix = interpolate( interp, ( float( x ) - at_l ) / ( at_r - at_l ) );    // :344 interp( ... ), Interpolator.cpp:14-101
				const float share_l = ( 1.0f - mix ) * mf_l.m;
				const float share_r = mix * mf_r.m;
				const float weight = share_l + share_r;
				const float freq_by_weight = share_l * mf_l.f + share_r * mf_r.f;
				if( weight == 0.0f ) break;                                    // :350-351 (`return` leaves this frame pair)
				const MFd o = { 0.0f, 0.0f };                                       // the cleared output MF this pair alone reaches
				op[int64_t( x ) * bins] = MFd{ o.m + weight, ( o.f * o.m + freq_by_weight ) / ( o.m + weight ) };   // :354-355
				}
			clear( x, x1 );                                                         // frames the pair left untouched
			}
		else
			{
			// :340-342 walks x from start_frame to end_frame and skips the frames outside [0, Fo) before computing anything: the
			// same frames, in the same order, without spinning through up to 2^31 skipped ones when a map value is huge
			const int step = forward ? 1 : -1;
			const int x_first = forward ? max( start_frame, 0 ) : min( start_frame, Fo32 - 1 );
			const int x_stop  = forward ? min( end_frame, Fo32 ) : max( end_frame, -1 );
			for( int x = x_first; forward ? x < x_stop : x > x_stop; x += step )
				{
				const float mix = interpolate( interp, ( float( x ) - at_l ) / ( at_r - at_l ) );    // :344 interp( ... ), Interpolator.cpp:14-101
				const float share_l = ( 1.0f - mix ) * mf_l.m;
				const float share_r = mix * mf_r.m;
				const float weight = share_l + share_r;
				const float freq_by_weight = share_l * mf_l.f + share_r * mf_r.f;
				if( weight == 0.0f ) break;                                    // :350-351
				MFd o = op[int64_t( x ) * bins];
				o.f = ( o.f * o.m + freq_by_weight ) / ( o.m + weight );      // :354
				o.m += weight;                                                 // :355
				op[int64_t( x ) * bins] = o;
				}
			}
		mf_l = mf_r; at_l = at_r;
		}
	if( !sequential && f1 == F ) clear( min( max( int( ceilf( at_l ) ), 0 ), Fo32 ), Fo32 );   // beyond the last pair's interval
	}

// The same algorithm cut by OUTPUT chains, for time maps that never run backwards (every column monotone): one thread owns the
// output frames [c L, (c+1) L) of one (channel, bin) column, c = a chain of the convert_to_audio that will follow.  It finds the
// first frame pair reaching into its range by bisection on the monotone map, writes every frame of the range exactly once
// (values, or the cleared { 0, 0 }) and -- having the chain's frequencies in hand, in frame order -- also leaves that chain's
// phase-increment sum (k_phase_sums2's result, bit for bit) and the NaN flag in the synthesis workspace.  A frame pair that
// straddles two chains is evaluated by both owners up to their own frames, so the reference's early exit (:350-351) is seen by both.
struct TimeChainParams
	{
	const MFd * in; const float * mod; MFd * out;
	int64_t F, Fo;
	int num_channels, bins, L, chains_per_channel;
	float sr, hop, analysis_rate;
	DivC ar_div;              // analysis_rate as a divisor (pv_math.h): the term of the phase sums, as k_phase_sums2 forms it
	double * sums;            // [ch][chains][bins]
	int * words;              // workspace tail: [0] NaN flag, [2] epoch, [4] "sums valid" (set iff equal to the epoch)
	int epoch;
	int * nonmono;            // [bins + 1], NOT cleared: a flag is set when it holds flag_tag, a number no earlier call used (core.hip: next_epoch); written by the launch's first flag_blocks blocks
	int flag_tag;
	int flag_blocks;          // a multiple of 8
	int interp;               // FLANHIP_INTERP_*: the Interpolator of PVModify.cpp:344 (0 = linear)
	};

// SUMS = false: the plain flanhip_modify_time_dev uses the same cut (any chain length) without a workspace.
//
// Round 4 rebuilt the kernel around what the compiled round-1 version turned out to do (profiles/r04_config3_*).  Its output loop ended, every
// iteration, in `s_waitcnt vmcnt(0)`: interpolate()'s table arm holds a load, the arms join below it, and on gfx950 loads and stores retire
// through ONE in-order counter -- a wavefront waited for the acknowledgement of the frame it had just stored before computing the next
// (one 512-byte store in flight per wavefront).  Now
//   * LINEAR (the Interpolator every stretch uses) is a compile-time arm: no load anywhere in the output loop, stores leave back to back;
//   * input frames arrive B at a time (MF + map value: 12 bytes per frame and lane) and their pairs are worked off from registers: the wait
//     for a batch drains the previous batch's stores -- one round trip per B input frames instead of one per output frame (an in-order
//     counter cannot wait for a load without the stores issued before it);
//   * threads are numbered ( chain, channel, bin ): with the XCD-contiguous block order the channels of a chain -- which read the same rows
//     of the time map -- meet in one L2 (channel-major numbering gave each XCD a channel of its own for the bench shape: the 23 MB map went
//     through the fabric eight times, 184 of the launch's 681 MB of L2 misses; 267 -> 235 us);
//   * the output pointer steps by a row per frame (every frame of [x_lo, x_hi) is written exactly once, in order); plain stores, NOT
//     non-temporal ones: a row piece shares its first and last 128-byte line with the neighbouring wavefronts' pieces (rows are 8200 bytes),
//     and those partial lines have to meet in the L2 before they leave for memory (tools/ubench/mtc_patterns.hip: stores alone 4.9 TB/s
//     plain, 3.65 non-temporal; whole rows per block, two bins per thread, other block orders and chain lengths: all slower or equal);
//   * the term of the phase sums divides by the analysis rate with the proven 3-instruction quotient (k_phase_sums2's own div_c);
//   * the map's monotonicity check (k_time_map_flags' job) runs INSIDE this launch, on its first `flag_blocks` blocks, while the chain blocks
//     work on the assumption that the map is monotone (every stretch).  k_modify_time, launched behind, reads the complete flags: if a
//     column does run backwards it redoes the whole PV and takes the sums' validity word back.  (Chain blocks that already see the flag
//     retire early; one that does not writes frames k_modify_time overwrites: every access is clamped, every loop bounded.)
// Results: bit for bit the round-1 kernel's (tests/test_gpu_processors.py, test_gpu_full_size.py::test_config3_*).
constexpr int kFlagRun = 16;                                                       // frame pairs per thread of the flag blocks
template<bool SUMS, bool LINEAR, int B>
__global__ __launch_bounds__( 256 ) void k_modify_time_chains( TimeChainParams p )
	{
	if( int( blockIdx.x ) < p.flag_blocks )
		{
		// nonmono[bin] = flag_tag for columns whose map runs backwards somewhere (or is NaN), nonmono[bins] = flag_tag if there is any: thread = ( run of
		// kFlagRun frame pairs, bin ), lanes along the bins
		const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
		const int64_t runs = ( p.F - 1 + kFlagRun - 1 ) / kFlagRun;
		if( idx >= runs * p.bins ) return;
		const int bin = int( idx % p.bins );
		const int64_t f0 = ( idx / p.bins ) * kFlagRun;                              // pairs ( f0, f0+1 ) ... ( f0+kFlagRun-1, f0+kFlagRun )
		float t[kFlagRun + 1];
		#pragma unroll
		for( int i = 0; i <= kFlagRun; ++i ) t[i] = p.mod[min( f0 + i, p.F - 1 ) * p.bins + bin];
		bool backwards = false;
		#pragma unroll
		for( int i = 0; i < kFlagRun; ++i )
			backwards |= !( time_to_frame( t[i + 1], p.sr, p.hop ) >= time_to_frame( t[i], p.sr, p.hop ) );   // :332 (r >= l; false for NaN)
		if( backwards ) { p.nonmono[bin] = p.flag_tag; p.nonmono[p.bins] = p.flag_tag; }
		return;
		}
	if( p.nonmono[p.bins] == p.flag_tag ) return;                                   // already known to run backwards somewhere: k_modify_time does this PV
	const int cblock = int( blockIdx.x ) - p.flag_blocks;                           // (flag_blocks is a multiple of 8: the XCD of a block is cblock % 8 too)
	const int64_t nblocks = int64_t( gridDim.x ) - p.flag_blocks, per_xcd = ( nblocks + 7 ) / 8;   // XCD x takes a contiguous run of blocks (pieces of a row meet in one L2)
	const int64_t vblock = int64_t( cblock % 8 ) * per_xcd + cblock / 8;
	const int64_t idx = vblock * blockDim.x + threadIdx.x;
	// index order ( chain, channel, bin ): the channels of a chain sit in neighbouring blocks -- one XCD, one L2 -- because they all read the
	// same rows of the time map (channel-major order sent each channel to an XCD of its own for the bench shape, and the map through the
	// fabric eight times: 184 of the launch's 681 MB of L2 misses)
	const int64_t per_chain = int64_t( p.num_channels ) * p.bins;
	const bool live = int64_t( cblock / 8 ) < per_xcd && idx < per_chain * p.chains_per_channel;
	unsigned worst = 0;                                                             // largest |bit pattern| among the written m and f: Inf / NaN sort on top
	if( live )
		{
		const int chain = int( idx / per_chain ), channel = int( ( idx % per_chain ) / p.bins ), bin = int( idx % p.bins );
		const int bins = p.bins;
		const int Fo32 = int( p.Fo );
		const int x_lo = chain * p.L, x_hi = min( x_lo + p.L, Fo32 );
		const cf * ip = reinterpret_cast<const cf*>( p.in ) + int64_t( channel ) * p.F * bins + bin;
		const float * mp = p.mod + bin;
		auto frame_of = [&]( int64_t k ) { return time_to_frame( mp[k * bins], p.sr, p.hop ); };
		auto clamped_ceil = [&]( float at ) { return min( max( int( ceilf( at ) ), 0 ), Fo32 ); };
		auto ends_beyond = [&]( int64_t k ) { return clamped_ceil( frame_of( k ) ) > x_lo; };   // pair ( k-1, k ); non-decreasing in k

		// The first pair that ends beyond x_lo: the answer is in [1, F] (F = none).  A plain bisection is 13 dependent reads of the map before the
		// first byte moves; a map that is anywhere near proportional (every constant stretch) is found in two or three by galloping away from
		// the proportional guess, the bisection finishing what is left.
		int64_t lo = 1, hi = p.F;
		if( p.F > 2 )
			{
			const int64_t g = min( max( int64_t( x_lo ) * ( p.F - 1 ) / p.Fo, int64_t( 1 ) ), p.F - 1 );
			int64_t w = 1;
			if( ends_beyond( g ) )
				{
				hi = g;                                                                  // answer in [1, g]
				while( true )
					{
					const int64_t q = hi - w;
					if( q < 1 ) break;
					if( ends_beyond( q ) ) { hi = q; w *= 2; } else { lo = q + 1; break; }
					}
				}
			else
				{
				lo = g + 1;                                                              // answer in [g + 1, F]
				while( true )
					{
					const int64_t q = lo + w - 1;
					if( q >= p.F ) break;
					if( !ends_beyond( q ) ) { lo = q + 1; w *= 2; } else { hi = q; break; }
					}
				}
			}
		while( lo < hi )
			{
			const int64_t mid = ( lo + hi ) >> 1;
			if( ends_beyond( mid ) ) hi = mid; else lo = mid + 1;
			}
		double ph = 0.0;
		int summed = 0;
		const int full8 = ( x_hi - x_lo ) & ~7;
		cf * o = reinterpret_cast<cf*>( p.out ) + ( int64_t( channel ) * p.Fo + x_lo ) * bins + bin;   // the next frame to write
		int cursor = x_lo;                                                          // ... and its number
		auto add_frame = [&]( float m, float f )                                    // every frame of [x_lo, x_hi) exactly once, ascending
			{
			*o = cf{ m, f };
			o += bins; ++cursor;
			if constexpr( SUMS )
				{
				worst = max( worst, max( __float_as_uint( m ) & 0x7FFFFFFFu, __float_as_uint( f ) & 0x7FFFFFFFu ) );
				ph += double( div_c( f, p.ar_div ) * FLANHIP_PI2_F );               // phase_vocoder.cpp:57 (k_phase_sums2's term)
				++summed;
				// k_phase_sums2 keeps its partial sum small after every full group of 8 frames of the chain
				if( ( summed & 7 ) == 0 && summed <= full8 && !( fabs( ph ) < 1.0e8 ) ) ph = fold_phase_any( ph );
				}
			};
		if( lo < p.F )
			{
			float at_l = frame_of( lo - 1 );
			cf mf_l = ip[( lo - 1 ) * bins];
			bool stop = false;
			for( int64_t k = lo; k < p.F && !stop; k += B )                         // :328, B pairs per trip
				{
				cf mf[B]; float tm[B];
				#pragma unroll
				for( int j = 0; j < B; ++j )
					{
					const int64_t kk = min( k + j, p.F - 1 ) * bins;
					mf[j] = ip[kk]; tm[j] = mp[kk];
					}
				// every map value of the batch is converted here, last one first: the one wait for the whole batch (which also drains the previous
				// batch's stores) stands in front of the pairs, and nothing below it waits for memory again
				#pragma unroll
				for( int j = B - 1; j >= 0; --j )
					{
					asm volatile( "" : "+v"( tm[j] ), "+v"( mf[j] ) );                    // (the values are wanted HERE, whichever pairs turn out to be walked)
					tm[j] = time_to_frame( tm[j], p.sr, p.hop );                        // :331
					}
				#pragma unroll
				for( int j = 0; j < B; ++j )
					{
					if( stop || k + j >= p.F ) continue;
					const cf mf_r = mf[j];
					const float at_r = tm[j];
					const int xs = clamped_ceil( at_l ), xe = clamped_ceil( at_r );     // :334-335, :342 (monotone: forward or empty)
					if( xs >= x_hi ) { stop = true; continue; }
					const int a = max( xs, x_lo ), b = min( xe, x_hi );
					while( cursor < a ) add_frame( 0.0f, 0.0f );                         // frames no pair reaches (before the first pair)
					bool left = false;                                                  // has the reference left this pair (:350-351)?
					const float span = at_r - at_l;
					for( int x = xs; x < b; ++x )
						{
						float m = 0.0f, f = 0.0f;
						if( !left )
							{
							const float pos = ( float( x ) - at_l ) / span;
							const float mix = LINEAR ? pos : interpolate( p.interp, pos );  // :344
							const float share_l = ( 1.0f - mix ) * mf_l.x;
							const float share_r = mix * mf_r.x;
							const float weight = share_l + share_r;
							const float freq_by_weight = share_l * mf_l.y + share_r * mf_r.y;
							if( weight == 0.0f ) left = true;
							else
								{
								const float om = 0.0f, of = 0.0f;                               // the cleared output MF this pair alone reaches
								m = om + weight; f = ( of * om + freq_by_weight ) / ( om + weight );   // :354-355
								}
							}
						if( x == cursor ) add_frame( m, f );                                // x >= a, for a monotone map; a store never lands outside [x_lo, x_hi) whatever the map
						}
					mf_l = mf_r; at_l = at_r;
					}
				}
			}
		while( cursor < x_hi ) add_frame( 0.0f, 0.0f );                             // beyond the last pair
		if constexpr( SUMS )
			p.sums[( int64_t( channel ) * p.chains_per_channel + chain ) * bins + bin] =
				( fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph );
		}
	if constexpr( SUMS )
		{
		const bool any_bad = __any( worst > 0x7F7FFFFFu );
		if( ( threadIdx.x & 63 ) == 0 && any_bad ) p.words[0] = p.epoch;
		if( cblock == 0 && threadIdx.x == 0 ) { p.words[2] = p.epoch; p.words[4] = p.epoch; }   // (k_modify_time takes [4] back if the map runs backwards)
		}
	}

// PV::stretch, PVModify.cpp:376-382: inclusive running sum over frames per bin in fp32 (sequential order = the
// reference's rounding), then frame_to_time; also the maximum of the result (FunctionSample::maximum, :312).
// A block owns 64 bins (column_scan, processors_common.h).
constexpr int kMapTB = 16, kMapTF = 224, kMapThreads = 512;                        // k_stretch_map tile: 16 bins x 224 frames, three of them in flight (44 KB of LDS); one scanning and seven moving waves (fifteen: no faster)
// IDX32: F * bins fits 32 bits with room (every map up to hours of audio): a row's address is the grid pointer plus a 32-bit offset
template<bool IDX32>
__global__ __launch_bounds__( kMapThreads ) void k_stretch_map( float * factor, int64_t F, int bins, float sr, float hop, float * d_max )
	{
	__shared__ __attribute__(( aligned( 16 ) )) float lds[3 * column_scan_lds_floats( kMapTF, kMapTB, 1 )];
	const int strip = xcd_contiguous_strip( blockIdx.x, ( bins + kMapTB - 1 ) / kMapTB );
	if( strip < 0 ) return;
	// (columns past the last bin of the last strip move and scan the last bin's column once more: the same values to the same addresses)
	const int bin = min( strip * kMapTB + int( threadIdx.x % kMapTB ), bins - 1 );
	auto at = [&]( int64_t f ) -> float *
		{
		if constexpr( IDX32 ) return factor + unsigned( int( f ) * bins + bin );
		else return factor + ( f * bins + bin );
		};
	float run = -0.0f, mx = -INFINITY;                                             // -0 + x == x for every x: frame 0 needs no special case
	column_scan_piped<kMapTF, kMapTB, kMapThreads>( lds, F, at,
		[&]( float v ) { run = v + run; return run; },                               // factor[frame] += factor[frame-1]
		[&]( int64_t, float v )
			{
			const float t = frame_to_time( v, sr, hop );                              // PVModify.cpp:381-382
			mx = fmaxf( mx, t );
			return t;
			} );
	// the block's maximum: over the lanes of each wave, then over the 4 waves through LDS, then into *d_max
	for( int o = 32; o > 0; o >>= 1 ) mx = fmaxf( mx, __shfl_xor( mx, o ) );
	__shared__ float wmax[kMapThreads / 64];
	if( ( threadIdx.x & 63 ) == 0 ) wmax[threadIdx.x >> 6] = mx;
	__syncthreads();
	if( threadIdx.x == 0 )
		{
		for( int w = 1; w < kMapThreads / 64; ++w ) mx = fmaxf( mx, wmax[w] );      // (thread 0's own mx is wmax[0])
		if( d_max && mx > -INFINITY )
			{
			int * addr = reinterpret_cast<int*>( d_max );
			int old = *addr;
			while( true )
				{
				const float cur = __int_as_float( old );
				if( !( mx > cur ) ) break;
				const int prev = atomicCAS( addr, old, __float_as_int( mx ) );
				if( prev == old ) break;
				old = prev;
				}
			}
		}
	}

// PV::stretch with a CONSTANT factor (stretch by 2: the everyday call, and what a callable returning a constant samples to): every column of the
// map is the same running sum of one number, and that sum has a closed form (const_sum.h: a few dozen steps for any frame, bit for bit the
// sequential additions) -- no grid to fill, nothing to scan: a block computes the values of its kConstRows frames (one thread each) and all its
// threads write the rows.  The maximum of a monotone sequence sits at one of its ends: block 0 writes it, no atomics and nothing to initialise.
constexpr int kConstRows = 16;
// LOG2K >= 0: the host has walked the sum once and hands over its value at every 2^LOG2K-th frame (const_sum.h); -1: every block walks for itself
// (sums whose runs do not fit: none seen).  K is a template parameter and the marks are the LAST argument so that the address of a block's mark
// depends on nothing that is itself loaded: every kernel-argument read goes out at once (they sit behind a ~2 us path: as a chain of four dependent
// reads would queue up).  Measured (tools/time_const_map.py): 12.7 us for 5626 x 1025 -- ~6 us of launch with a 3.9 KB argument and prologue, ~6.5 us of
// stores -- against 54.5 us for fill + scan.
template<int LOG2K>
__global__ __launch_bounds__( 256 ) void k_stretch_map_const( float * map, int64_t F, int bins, float sr, float hop, float * d_max, ConstSumMarks marks )
	{
	__shared__ float row_value[kConstRows];
	const int64_t f0 = int64_t( blockIdx.x ) * kConstRows;
	const int rows = int( min( int64_t( kConstRows ), F - f0 ) );
	// frame t's sum from the mark at or below it: fewer than K real additions -- the recurrence itself (PVModify.cpp:376-378)
	auto sum_at = [&]( int64_t t, int lane_steps )
		{
		if constexpr( LOG2K < 0 ) return const_running_sum( marks.c, uint64_t( t + lane_steps ) );
		else
			{
			const int64_t k = t >> LOG2K;
			float v = marks.at[k];
			const int steps = int( t - ( k << LOG2K ) ) + lane_steps;
			const int most = int( t - ( k << LOG2K ) ) + kConstRows - 1;             // (wave-uniform bound)
			for( int j = 0; j < most; ++j ) v = ( j < steps ) ? v + marks.c : v;
			return v;
			}
		};
	if( threadIdx.x < 64 )
		{
		const float v = sum_at( f0, min( int( threadIdx.x ), kConstRows - 1 ) );
		if( int( threadIdx.x ) < rows ) row_value[threadIdx.x] = frame_to_time( v, sr, hop );                      // :381-382
		}
	if( blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128 && d_max )
		{
		// FunctionSample::maximum (:312) as the scanning kernel forms it: fmaxf over every element from -inf (a NaN never wins); the sequence is monotone
		const float a = frame_to_time( marks.c, sr, hop ), b = frame_to_time( sum_at( F - 1, 0 ), sr, hop );
		if( threadIdx.x == 64 ) *d_max = fmaxf( fmaxf( -INFINITY, a ), b );
		}
	__syncthreads();
	// the block's rows are one contiguous run of rows * bins floats: 16-byte stores between a scalar head and tail (a row pitch of 1025 floats
	// leaves a row 4-byte aligned only); the row of element i by a multiplication (i < 2^23: the float quotient is off by one at most)
	float * out = map + f0 * bins;
	const int total = rows * bins;
	const float inv_bins = 1.0f / float( bins );
	auto row_of = [&]( int i ) { int r = int( float( i ) * inv_bins ); r -= ( r * bins > i ); r += ( ( r + 1 ) * bins <= i ); return r; };
	const int head = min( total, int( ( 16u - unsigned( reinterpret_cast<uintptr_t>( out ) & 15u ) ) & 15u ) / 4 );
	const int quads = ( total - head ) / 4;
	if( int( threadIdx.x ) < head ) out[threadIdx.x] = row_value[0];
	for( int k = threadIdx.x; k < quads; k += 256 )
		{
		const int i = head + 4 * k;
		const int r = row_of( i ), left = ( r + 1 ) * bins - i;                       // elements of row r from i on
		const float a = row_value[r], b = row_value[min( r + 1, rows - 1 )];
		*reinterpret_cast<float4*>( out + i ) = make_float4( a, left > 1 ? a : b, left > 2 ? a : b, left > 3 ? a : b );
		}
	for( int i = head + 4 * quads + threadIdx.x; i < total; i += 256 ) out[i] = row_value[rows - 1];
	}

// PV::repitch, PVModify.cpp:278-284: inclusive running sum over bins per frame (fp32, sequential), bin_to_frequency.
// A block owns kRowsPerBlock frames: their rows are staged in LDS with coalesced loads, one thread per row carries the running
// sum along its row (32 values at a time through registers), all threads apply bin_to_frequency on the way out.
// s_rows: dynamic LDS, kRowsPerBlock * (bins + 1) floats.
constexpr int kRowsPerBlock = 8;
__global__ __launch_bounds__( 256 ) void k_repitch_scan( float * factor, int64_t F, int bins, float sr, float dft )
	{
	extern __shared__ float s_scan_rows[];
	const int stride = bins + 1;                                                    // rows start in different banks
	const int64_t frame0 = int64_t( blockIdx.x ) * kRowsPerBlock;
	const int rows = int( min( int64_t( kRowsPerBlock ), F - frame0 ) );
	float * g = factor + frame0 * bins;
	for( int i = threadIdx.x; i < rows * bins; i += 256 ) s_scan_rows[( i / bins ) * stride + i % bins] = g[i];
	__syncthreads();
	if( threadIdx.x < rows )
		{
		float * r = s_scan_rows + threadIdx.x * stride;
		float run = -0.0f;                                                          // -0 + x == x: bin 0 needs no special case
		int b0 = 0;
		for( ; b0 + 32 <= bins; b0 += 32 )
			{
			float v[32];
			#pragma unroll
			for( int j = 0; j < 32; ++j ) v[j] = r[b0 + j];
			#pragma unroll
			for( int j = 0; j < 32; ++j ) { run = v[j] + run; v[j] = run; }         // factor[frame][bin] += factor[frame][bin-1]
			#pragma unroll
			for( int j = 0; j < 32; ++j ) r[b0 + j] = v[j];
			}
		for( ; b0 < bins; ++b0 ) { run = r[b0] + run; r[b0] = run; }
		}
	__syncthreads();
	for( int i = threadIdx.x; i < rows * bins; i += 256 )
		g[i] = bin_to_frequency( s_scan_rows[( i / bins ) * stride + i % bins], sr, dft );   // :283-284
	}

// PVModify.cpp:289-302: every MF's own frequency looked up (lerp) in the per-frame map.
__global__ __launch_bounds__( 256 ) void k_repitch_lerp( const MFd * in, int64_t count, int64_t F, int bins, float sr, float dft,
	const float * map, float * in_modified )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= count ) return;
	const int64_t frame = ( idx / bins ) % F;
	const float fbin = fminf( fmaxf( frequency_to_bin( in[idx].f, sr, dft ), 0.0f ), float( bins - 1 ) - 0.0001f ); // std::clamp
	const int lo = int( floorf( fbin ) );
	const int hi = lo + 1;
	const float lo_freq = map[frame * bins + lo];
	const float hi_freq = map[frame * bins + hi];
	const float r = fbin - float( lo );
	in_modified[idx] = lo_freq * ( 1.0f - r ) + hi_freq * r;
	}

// modify_frequency_base, PVModify.cpp:207-253.  One WAVEFRONT per (channel, frame) row; the output row is assembled in LDS
// (cleared there, :205) and leaves with coalesced stores: no separate clearing pass over HBM and no read-modify-write.
// Same argument as k_modify_time: when the bin map of a row never runs backwards (hiBin >= loBin for every adjacent pair --
// any repitch by positive factors) the output intervals [ceil(lo), ceil(hi)) of the pairs are disjoint, each output bin is
// touched at most once and the pairs of the row are independent: the lanes take them 64 at a time.  A row whose map
// does run backwards is walked by lane 0 in the reference order.  s_rows: dynamic LDS, bins MF per wave.
// REPITCH: in_modified is not read but evaluated on the fly, PV::repitch's own rule (PVModify.cpp:289-302: every MF's frequency
// looked up, lerp, in the row's map); the map row is then staged in LDS behind the output rows (bins floats per wave).
// LINEAR: the linear Interpolator (every default call) as a compile-time arm -- interpolate()'s table arm holds a flat load, and where the arms join
// every iteration of the pair loop waited for the whole memory queue.
// Memory reads come in BATCHES of kFreqBatch loop steps, requested together ahead of the steps' arithmetic: with a load and its wait inside every
// step a row cost its wavefront 17 + 17 memory round trips one after the other (0.35 ms for 45 008 rows whose bytes take 0.15; staging the input
// row in LDS instead was measured: it halves the resident wavefronts and the kernel took 0.76 ms).
constexpr int kFreqBatch = 6;
template<bool REPITCH, bool LINEAR>
__global__ __launch_bounds__( 256 ) void k_modify_frequency( const MFd * in, int num_channels, int64_t F, int bins, float sr, float dft,
	const float * mod, const float * in_modified, MFd * out, int interp )
	{
	extern __shared__ MFd s_rows[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t idx = int64_t( blockIdx.x ) * ( blockDim.x >> 6 ) + wave;
	if( idx >= int64_t( num_channels ) * F ) return;
	const int64_t frame = idx % F;
	const MFd * row = in + idx * bins;
	MFd * orow = s_rows + size_t( wave ) * bins;
	const float * mrow = mod + frame * bins;
	const float * irow = REPITCH ? nullptr : in_modified + idx * bins;
	constexpr int B = kFreqBatch;
	bool backwards = false;
	if constexpr( REPITCH )
		{
		float * smap = reinterpret_cast<float*>( s_rows + size_t( blockDim.x >> 6 ) * bins ) + size_t( wave ) * bins;
		for( int bin0 = lane; bin0 < bins; bin0 += 64 * B )
			{
			float v[B];
			#pragma unroll
			for( int j = 0; j < B; ++j ) v[j] = mrow[min( bin0 + 64 * j, bins - 1 )];
			#pragma unroll
			for( int j = 0; j < B; ++j ) { const int bin = bin0 + 64 * j; if( bin < bins ) { smap[bin] = v[j]; orow[bin] = MFd{ 0.0f, 0.0f }; } }   // :205 clear_buffer
			}
		mrow = smap;
		wave_sync();
		for( int bin = 1 + lane; bin < bins; bin += 64 ) backwards |= !( frequency_to_bin( mrow[bin], sr, dft ) >= frequency_to_bin( mrow[bin - 1], sr, dft ) );
		}
	else
		{
		for( int bin0 = lane; bin0 < bins; bin0 += 64 * B )
			{
			float v[B], u[B];
			#pragma unroll
			for( int j = 0; j < B; ++j ) { const int bin = min( bin0 + 64 * j, bins - 1 ); v[j] = mrow[bin]; u[j] = mrow[max( bin - 1, 0 )]; }
			#pragma unroll
			for( int j = 0; j < B; ++j )
				{
				const int bin = bin0 + 64 * j;
				if( bin >= bins ) continue;
				orow[bin] = MFd{ 0.0f, 0.0f };                                          // :205 clear_buffer
				if( bin >= 1 ) backwards |= !( frequency_to_bin( v[j], sr, dft ) >= frequency_to_bin( u[j], sr, dft ) );
				}
			}
		}
	const bool sequential = __any( backwards );
	wave_sync();

	auto modified_of = [&]( float f_of_bin ) -> float                                // REPITCH: the frequency an MF of frequency f is moved to (PVModify.cpp:289-302)
		{
		const float fbin = fminf( fmaxf( frequency_to_bin( f_of_bin, sr, dft ), 0.0f ), float( bins - 1 ) - 0.0001f );   // :293 std::clamp
		const int lo = int( floorf( fbin ) );
		const float r = fbin - float( lo );
		return mrow[lo] * ( 1.0f - r ) + mrow[lo + 1] * r;                         // :295-301 (mrow: the LDS copy)
		};
	// :214-244 for the pair ( bin-1, bin ), its operands in hand: the two MFs, their map entries (as bins), the frequencies they are moved to
	auto pair = [&]( MFd in_lo, MFd in_hi, float map_lo, float map_hi, float to_lo, float to_hi )
		{
		const float loBin = frequency_to_bin( map_lo, sr, dft );                    // :218
		const float hiBin = frequency_to_bin( map_hi, sr, dft );                    // :219
		const MFd loMF = { in_lo.m, to_lo };                                        // :227
		const MFd hiMF = { in_hi.m, to_hi };                                        // :228
		const bool forward = hiBin > loBin;                                         // :220
		const int loR = int( forward ? ceilf( loBin ) : floorf( loBin ) );          // :222
		const int hiR = int( forward ? ceilf( hiBin ) : floorf( hiBin ) );          // :223
		const int start_bin = min( max( loR, 0 ), bins - 1 );                       // :224
		const int end_bin   = min( max( hiR, 0 ), bins - 1 );                       // :225
		for( int y = start_bin; y != end_bin; forward ? ++y : --y )                 // :230
			{
			const float xq = ( float( y ) - loBin ) / ( hiBin - loBin );
			const float mix = LINEAR ? xq : interpolate( interp, xq );              // :232 interp( ... )
			const float share_l = ( 1.0f - mix ) * loMF.m;
			const float share_r = mix * hiMF.m;
			const MFd mx = share_l < share_r ? loMF : hiMF;                         // :237
			MFd o = orow[y];
			if( mx.m > o.m )                                                        // :239
				{
				o.m += mx.m;                                                        // :241
				o.f = mx.f;                                                         // :242
				orow[y] = o;
				}
			}
		};
	auto pair_at = [&]( int bin )                                                   // (the sequential walk: one lane, operands read as it goes)
		{
		const MFd a = row[bin - 1], b = row[bin];
		if constexpr( REPITCH ) pair( a, b, mrow[bin - 1], mrow[bin], modified_of( a.f ), modified_of( b.f ) );
		else pair( a, b, mrow[bin - 1], mrow[bin], irow[bin - 1], irow[bin] );
		};
	if( !sequential )
		{
		for( int bin0 = 1 + lane; bin0 < bins; bin0 += 64 * B )
			{
			MFd a[B], b[B]; float ml[B], mh[B], tl[B], th[B];
			#pragma unroll
			for( int j = 0; j < B; ++j )
				{
				const int bin = min( bin0 + 64 * j, bins - 1 );
				a[j] = row[bin - 1]; b[j] = row[bin];
				if constexpr( !REPITCH ) { ml[j] = mrow[bin - 1]; mh[j] = mrow[bin]; tl[j] = irow[bin - 1]; th[j] = irow[bin]; }
				}
			#pragma unroll
			for( int j = 0; j < B; ++j )
				{
				const int bin = bin0 + 64 * j;
				if( bin >= bins ) continue;
				if constexpr( REPITCH ) pair( a[j], b[j], mrow[bin - 1], mrow[bin], modified_of( a[j].f ), modified_of( b[j].f ) );
				else pair( a[j], b[j], ml[j], mh[j], tl[j], th[j] );
				}
			}
		}
	else if( lane == 0 )
		{
		for( int bin = 1; bin < bins; ++bin ) pair_at( bin );
		}
	wave_sync();
	MFd * grow = out + idx * bins;
	for( int bin = lane; bin < bins; bin += 64 ) grow[bin] = orow[bin];
	}

// PV::shape, PV.cpp:431-454.  AFFINE: shaped = { a*m + b, c*f + d }; otherwise shaped values come from `shaped_tbl`.
template<bool AFFINE>
__global__ __launch_bounds__( 256 ) void k_shape_plain( const MFd * in, const MFd * shaped_tbl, int64_t count, float a, float b, float c, float d, MFd * out )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= count ) return;
	if( AFFINE ) { const MFd v = in[idx]; out[idx] = MFd{ a * v.m + b, c * v.f + d }; }   // :436, :452
	else out[idx] = shaped_tbl[idx];
	}

// The same without alignment, walking chains: one thread per ( channel, chain of L frames, bin ) -- lanes are adjacent bins, so every
// step moves whole coalesced row segments -- writes its frames in order and leaves convert_to_audio's pre-pass for the new PV in the
// synthesis workspace: the chain's phase sum, bit for bit what k_phase_sums2 would compute from the output, and the NaN / Inf flag.
struct ShapeChainParams
	{
	const MFd * in; const MFd * tbl; MFd * out;
	int64_t F;
	int num_channels, bins, L, chains_per_channel;
	float a, b, c, d, analysis_rate;
	double * sums;            // [ch][chains][bins]
	int * words;              // workspace tail: [0] NaN flag, [2] epoch, [4] "sums valid" (set iff equal to the epoch)
	int epoch;
	};

template<bool AFFINE>
__global__ __launch_bounds__( 256 ) void k_shape_plain_chains( ShapeChainParams p )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	const int64_t per_channel = int64_t( p.chains_per_channel ) * p.bins;
	bool bad = false;
	if( idx < per_channel * p.num_channels )
		{
		const int channel = int( idx / per_channel ), chain = int( ( idx % per_channel ) / p.bins ), bin = int( idx % p.bins );
		const int64_t x_lo = int64_t( chain ) * p.L, x_hi = min( x_lo + p.L, p.F );
		const int64_t base = int64_t( channel ) * p.F * p.bins + bin;
		double ph = 0.0;
		int summed = 0;
		const int groups = int( x_hi - x_lo ) & ~7;
		#pragma unroll 4
		for( int64_t x = x_lo; x < x_hi; ++x )
			{
			const int64_t at = base + x * p.bins;
			MFd v;
			if( AFFINE ) { const MFd u = p.in[at]; v = MFd{ p.a * u.m + p.b, p.c * u.f + p.d }; }   // PV.cpp:436, :452
			else v = p.tbl[at];
			p.out[at] = v;
			bad |= !( fabsf( v.m ) <= 3.4028235e38f ) || !( fabsf( v.f ) <= 3.4028235e38f );
			ph += double( v.f / p.analysis_rate * FLANHIP_PI2_F );                  // phase_vocoder.cpp:57 (k_phase_sums2's term)
			++summed;
			// k_phase_sums2 keeps its partial sum small after every full group of 8 frames of the chain
			if( ( summed & 7 ) == 0 && summed <= groups && !( fabs( ph ) < 1.0e8 ) ) ph = fold_phase_any( ph );
			}
		p.sums[( int64_t( channel ) * p.chains_per_channel + chain ) * p.bins + bin] =
			( fabs( ph ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( ph ) : fold_phase_any( ph );
		}
	const bool any_bad = __any( bad );
	if( ( threadIdx.x & 63 ) == 0 && any_bad ) p.words[0] = p.epoch;
	if( blockIdx.x == 0 && threadIdx.x == 0 ) { p.words[2] = p.epoch; p.words[4] = p.epoch; }
	}

// Shift alignment (PV.cpp:438-448): one WAVEFRONT per (channel, frame) row; the sequential "strictly louder replaces" rule is
// resolved through LDS keys (processors_common.h placement_offer).  keys: dynamic LDS, bins u64 per wave.
template<bool AFFINE>
__global__ __launch_bounds__( 256 ) void k_shape_aligned( const MFd * in, const MFd * shaped_tbl, int64_t rows, int bins, float sr, float dft,
	float a, float b, float c, float d, MFd * out )
	{
	extern __shared__ unsigned long long s_keys[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t idx = int64_t( blockIdx.x ) * ( blockDim.x >> 6 ) + wave;
	if( idx >= rows ) return;
	unsigned long long * keys = s_keys + size_t( wave ) * bins;
	const MFd * row = in + idx * bins;
	MFd * orow = out + idx * bins;
	// (memory reads in batches of kFreqBatch loop steps, requested together ahead of the steps' arithmetic: one round trip per batch, not per step)
	constexpr int B = kFreqBatch;
	auto shape_of = [&]( MFd v, MFd tbl ) { return AFFINE ? MFd{ a * v.m + b, c * v.f + d } : tbl; };   // :436
	for( int bin = lane; bin < bins; bin += 64 ) keys[bin] = 0ull;
	wave_sync();
	for( int bin0 = lane; bin0 < bins; bin0 += 64 * B )                             // :433
		{
		MFd v[B], t[B];
		#pragma unroll
		for( int j = 0; j < B; ++j )
			{
			const int bin = min( bin0 + 64 * j, bins - 1 );
			v[j] = row[bin];
			if constexpr( !AFFINE ) t[j] = shaped_tbl[idx * bins + bin]; else t[j] = MFd{ 0.0f, 0.0f };
			}
		#pragma unroll
		for( int j = 0; j < B; ++j )
			{
			const int bin = bin0 + 64 * j;
			if( bin >= bins ) continue;
			const MFd s = shape_of( v[j], t[j] );
			const int binShift = to_int_sat( float( bin ) - frequency_to_bin( v[j].f, sr, dft ) );          // :440
			const int shapedBin = to_int_sat( frequency_to_bin( s.f, sr, dft ) + float( binShift ) );       // :441
			if( shapedBin < 0 || bins <= shapedBin ) continue;                      // :442
			placement_offer( keys, shapedBin, s.m, bin );                           // :445-447
			}
		}
	wave_sync();
	for( int bin0 = lane; bin0 < bins; bin0 += 64 * B )
		{
		unsigned long long key[B]; MFd v[B], t[B];
		#pragma unroll
		for( int j = 0; j < B; ++j ) key[j] = keys[min( bin0 + 64 * j, bins - 1 )];
		#pragma unroll
		for( int j = 0; j < B; ++j )
			{
			const int w = key[j] ? placement_winner( key[j] ) : 0;
			v[j] = row[w];
			if constexpr( !AFFINE ) t[j] = shaped_tbl[idx * bins + w]; else t[j] = MFd{ 0.0f, 0.0f };
			}
		#pragma unroll
		for( int j = 0; j < B; ++j )
			{
			const int bin = bin0 + 64 * j;
			if( bin < bins ) orow[bin] = key[j] ? shape_of( v[j], t[j] ) : MFd{ 0.0f, 0.0f };    // :426 clear_buffer
			}
		}
	}

static int launch_repitch_scan( float * d_factor, int64_t F, int bins, float sr, float dft, hipStream_t s )
	{
	const size_t lds = sizeof( float ) * kRowsPerBlock * size_t( bins + 1 );
	FLANHIP_REQUIRE( lds <= 160 * 1024, FLANHIP_ERR_UNSUPPORTED, "more than 5119 bins" );
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_repitch_scan ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	hipLaunchKernelGGL( k_repitch_scan, dim3( (unsigned) ( ( F + kRowsPerBlock - 1 ) / kRowsPerBlock ) ), dim3( 256 ), lds, s, d_factor, F, bins, sr, dft );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

} // namespace flanhip

namespace flanhip { int processors_set_interp_lut( int slot, const float * d_table ) { return set_interp_lut_here( slot, d_table ); } }

using namespace flanhip;

// the monotone-map kernel: its first blocks check the map (nonmono: tagged flags, no clearing), the others walk the output chains
// the monotone-map kernel: its first blocks check the map (nonmono: tagged flags, no clearing), the others walk the output chains
template<bool SUMS>
static int launch_time_chains( TimeChainParams & p, hipStream_t s )
	{
	const int64_t flag_threads = ( ( p.F - 1 + kFlagRun - 1 ) / kFlagRun ) * p.bins;
	p.flag_blocks = int( ( ( flag_threads + 255 ) / 256 + 7 ) / 8 * 8 );
	const int64_t owners = int64_t( p.num_channels ) * p.chains_per_channel * p.bins;
	const dim3 grid( (unsigned) ( p.flag_blocks + ( ( owners + 255 ) / 256 + 7 ) / 8 * 8 ) ), block( 256 );
	// frames per trip: 2 with the linear Interpolator (58 VGPRs with the sums: eight wavefronts per SIMD; 4 frames: 68 registers, seven
	// wavefronts, 1 % slower), 4 where the Interpolator is a run-time choice (80 registers either way)
	if( p.interp != FLANHIP_INTERP_LINEAR ) hipLaunchKernelGGL( ( k_modify_time_chains<SUMS, false, 4> ), grid, block, 0, s, p );
	else hipLaunchKernelGGL( ( k_modify_time_chains<SUMS, true, 2> ), grid, block, 0, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

extern "C" {

static int modify_time_dev_impl( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int hop, const float * d_mod,
	int64_t Fo, flanhip_MF * d_out, int interp, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( valid_interp( interp ), FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( d_mod && hop >= 1 && Fo > 0, FLANHIP_ERR_INVALID_ARG, "bad map / output length" );
	hipStream_t s = (hipStream_t) stream;
	FLANHIP_REQUIRE( Fo < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "2^31 output frames or more" );
	int * d_flags = nullptr;                                                       // per bin: does the time map ever run backwards?
	// (not cleared -- the runtime's fill is a 5 us kernel of its own: a flag counts as set when it holds this call's tag, and whatever the fresh
	// allocation holds is at best an EARLIER call's tag)
	FLANHIP_CHECK( hipMallocAsync( reinterpret_cast<void**>( &d_flags ), sizeof( int ) * size_t( bins + 1 ), s ) );
	// monotone maps (every stretch): output chains of 32 frames, one thread per (channel, bin, chain) -- see k_modify_time_chains
	TimeChainParams cp{};
	cp.in = (const MFd*) d_pv; cp.mod = d_mod; cp.out = (MFd*) d_out;
	cp.F = F; cp.Fo = Fo; cp.num_channels = int( ch ); cp.bins = bins; cp.L = 32; cp.chains_per_channel = int( ( Fo + cp.L - 1 ) / cp.L );
	cp.sr = sr; cp.hop = float( hop ); cp.nonmono = d_flags; cp.flag_tag = next_epoch(); cp.interp = interp;
	if( int rc = launch_time_chains<false>( cp, s ) ) return rc;
	// the general walk, for a PV with a column that runs backwards (retires at once otherwise):
	// enough (column, segment) threads to fill the chip, segments of at least 16 frame pairs
	const int64_t columns = ch * bins;
	int segments = int( std::min<int64_t>( std::max<int64_t>( ( 256 * 2048 + columns - 1 ) / columns, 1 ), std::max<int64_t>( ( F - 1 ) / 16, 1 ) ) );
	const int64_t seg_len = std::max<int64_t>( ( F - 1 + segments - 1 ) / segments, 1 );
	const int64_t threads = columns * segments;
	hipLaunchKernelGGL( k_modify_time, dim3( (unsigned) ( ( threads + 255 ) / 256 ) ), dim3( 256 ), 0, s,
		(const MFd*) d_pv, int( ch ), F, bins, sr, float( hop ), d_mod, Fo, (MFd*) d_out, d_flags, cp.flag_tag, segments, seg_len, 1, interp, nullptr, 0 );
	FLANHIP_CHECK( hipGetLastError() );
	FLANHIP_CHECK( hipFreeAsync( d_flags, s ) );
	return FLANHIP_OK;
	}

static int modify_time_dev_fused_impl( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float analysis_rate, const float * d_mod,
	int64_t Fo, flanhip_MF * d_out, int window_size, void * d_ws, int interp, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( valid_interp( interp ), FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( d_mod && d_ws && analysis_rate > 0.0f && Fo > 0, FLANHIP_ERR_INVALID_ARG, "bad map / output length / workspace" );
	FLANHIP_REQUIRE( Fo < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "2^31 output frames or more" );
	hipStream_t s = (hipStream_t) stream;
	SynthLayout lay;                                                               // the chains convert_to_audio will cut the OUTPUT into
	if( int rc = synth_layout( ch, Fo, bins, sr, analysis_rate, window_size, &lay ) ) return rc;
	const int hop = lay.hop;                                                       // PVBuffer.cpp:381-384: what PV::modify_time works with
	int * d_flags = nullptr;                                                       // per bin: does the time map ever run backwards?  [bins]: any
	// in the workspace, not cleared: a flag counts as set when it holds this call's tag, and whatever the words hold is at best an EARLIER call's
	// (a stream-ordered allocation costs the stream ~6 us at its release, the runtime's fill is a 5 us kernel of its own)
	d_flags = reinterpret_cast<int*>( reinterpret_cast<char*>( d_ws ) + lay.flags_offset );
	TimeChainParams p{};
	p.in = (const MFd*) d_pv; p.mod = d_mod; p.out = (MFd*) d_out;
	p.F = F; p.Fo = Fo; p.num_channels = int( ch ); p.bins = bins; p.L = lay.L; p.chains_per_channel = lay.chains_per_channel;
	p.sr = sr; p.hop = float( hop ); p.analysis_rate = analysis_rate;
	DivPlan dp;
	if( int rc = get_div_plan( analysis_rate, &dp ) ) return rc;
	p.ar_div = DivC{ dp.c, dp.rc, dp.exact };
	p.sums = reinterpret_cast<double*>( d_ws );
	p.words = reinterpret_cast<int*>( reinterpret_cast<char*>( d_ws ) + lay.carry_bytes + lay.head_bytes );
	p.epoch = next_epoch();
	p.nonmono = d_flags; p.flag_tag = p.epoch; p.interp = interp;
	note_workspace_producer( d_ws, 0 );                                               // chain sums (maybe), no group totals: convert_to_audio runs its scan
	if( int rc = launch_time_chains<true>( p, s ) ) return rc;
	// the general walk, for a PV with a column that runs backwards: retires at once otherwise
	const int64_t columns = ch * bins;
	int segments = int( std::min<int64_t>( std::max<int64_t>( ( 256 * 2048 + columns - 1 ) / columns, 1 ), std::max<int64_t>( ( F - 1 ) / 16, 1 ) ) );
	const int64_t seg_len = std::max<int64_t>( ( F - 1 + segments - 1 ) / segments, 1 );
	const int64_t threads = columns * segments;
	hipLaunchKernelGGL( k_modify_time, dim3( (unsigned) ( ( threads + 255 ) / 256 ) ), dim3( 256 ), 0, s,
		(const MFd*) d_pv, int( ch ), F, bins, sr, float( hop ), d_mod, Fo, (MFd*) d_out, d_flags, p.flag_tag, segments, seg_len, 1, interp, p.words, p.epoch );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_modify_time_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int hop, const float * d_mod,
	int64_t Fo, flanhip_MF * d_out, void * stream )
	{
	return modify_time_dev_impl( d_pv, ch, F, bins, sr, hop, d_mod, Fo, d_out, FLANHIP_INTERP_LINEAR, stream );
	}

int flanhip_modify_time_interp_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int hop, const float * d_mod,
	int64_t Fo, int interp, flanhip_MF * d_out, void * stream )
	{
	return modify_time_dev_impl( d_pv, ch, F, bins, sr, hop, d_mod, Fo, d_out, interp, stream );
	}

int flanhip_modify_time_dev_fused( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float analysis_rate, const float * d_mod,
	int64_t Fo, flanhip_MF * d_out, int window_size, void * d_ws, void * stream )
	{
	return modify_time_dev_fused_impl( d_pv, ch, F, bins, sr, analysis_rate, d_mod, Fo, d_out, window_size, d_ws, FLANHIP_INTERP_LINEAR, stream );
	}

int flanhip_modify_time_interp_dev_fused( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float analysis_rate, const float * d_mod,
	int64_t Fo, int interp, flanhip_MF * d_out, int window_size, void * d_ws, void * stream )
	{
	return modify_time_dev_fused_impl( d_pv, ch, F, bins, sr, analysis_rate, d_mod, Fo, d_out, window_size, d_ws, interp, stream );
	}

int flanhip_modify_time( const flanhip_MF * pv, int64_t ch, int64_t F, int bins, float sr, int hop, const float * mod,
	int64_t Fo, flanhip_MF * out, volatile int * cancel )
	{
	if( int rc = check_pv_args( pv, out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( mod && hop >= 1 && Fo > 0, FLANHIP_ERR_INVALID_ARG, "bad map / output length" );
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	DevBuf d_pv, d_mod, d_out;
	const size_t in_bytes = sizeof( flanhip_MF ) * size_t( ch ) * F * bins, mod_bytes = sizeof( float ) * size_t( F ) * bins;
	const size_t out_bytes = sizeof( flanhip_MF ) * size_t( ch ) * Fo * bins;
	if( int rc = d_pv.alloc( in_bytes ) ) return rc;
	if( int rc = d_mod.alloc( mod_bytes ) ) return rc;
	if( int rc = d_out.alloc( out_bytes ) ) return rc;
	if( int rc_t = flanhip_upload( d_pv.p, pv, in_bytes ) ) return rc_t;
	if( int rc_t = flanhip_upload( d_mod.p, mod, mod_bytes ) ) return rc_t;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	if( int rc = flanhip_modify_time_dev( (const flanhip_MF*) d_pv.p, ch, F, bins, sr, hop, (const float*) d_mod.p, Fo, (flanhip_MF*) d_out.p, nullptr ) ) return rc;
	FLANHIP_CHECK( hipDeviceSynchronize() );
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	if( int rc_t = flanhip_download( out, d_out.p, out_bytes ) ) return rc_t;
	return FLANHIP_OK;
	}

int flanhip_stretch_map_dev( float * d_factor, int64_t F, int bins, float sr, int hop, float * d_max, void * stream )
	{
	FLANHIP_REQUIRE( d_factor && F > 0 && bins > 0 && hop >= 1, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	if( int rc = require_device() ) return rc;
	hipStream_t s = (hipStream_t) stream;
	if( d_max )
		{
		const float ninf = -INFINITY;
		FLANHIP_CHECK( hipMemcpyAsync( d_max, &ninf, sizeof( float ), hipMemcpyHostToDevice, s ) );
		}
	const dim3 grid( xcd_grid( ( bins + kMapTB - 1 ) / kMapTB ) );
	if( F * int64_t( bins ) < ( int64_t( 1 ) << 30 ) && !debug_options().wide_offsets ) hipLaunchKernelGGL( k_stretch_map<true>, grid, dim3( kMapThreads ), 0, s, d_factor, F, bins, sr, float( hop ), d_max );
	else hipLaunchKernelGGL( k_stretch_map<false>, grid, dim3( kMapThreads ), 0, s, d_factor, F, bins, sr, float( hop ), d_max );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

extern "C" int flanhip_stretch_map_const_dev( float factor, float * d_map, int64_t F, int bins, float sr, int hop, float * d_max, void * stream )
	{
	FLANHIP_REQUIRE( d_map && F > 0 && bins > 0 && hop >= 1, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	if( int rc = require_device() ) return rc;
	// (the closed-form kernel stores four floats at a time and assumes such a store spans at most two rows: rows of one or two bins take the scanning kernel)
	if( bins < 3 )
		{
		if( int rc = flanhip_fill_dev( d_map, F * bins, factor, stream ) ) return rc;
		return flanhip_stretch_map_dev( d_map, F, bins, sr, hop, d_max, stream );
		}
	const int64_t blocks = ( F + kConstRows - 1 ) / kConstRows;
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many frames for one launch" );
	static thread_local ConstSumMarks marks;
	const bool fits = const_sum_marks( factor, uint64_t( F - 1 ), &marks );
	if( !fits ) { marks.c = factor; marks.log2K = 0; marks.count = 0; }
	const dim3 grid( (unsigned) blocks ), block( 256 );
	hipStream_t st = (hipStream_t) stream;
	switch( fits ? marks.log2K : -1 )
		{
#define FLANHIP_CASE( L ) case L: hipLaunchKernelGGL( k_stretch_map_const<L>, grid, block, 0, st, d_map, F, bins, sr, float( hop ), d_max, marks ); break;
		FLANHIP_CASE( 4 ) FLANHIP_CASE( 5 ) FLANHIP_CASE( 6 ) FLANHIP_CASE( 7 ) FLANHIP_CASE( 8 ) FLANHIP_CASE( 9 ) FLANHIP_CASE( 10 ) FLANHIP_CASE( 11 ) FLANHIP_CASE( 12 )
#undef FLANHIP_CASE
		default: hipLaunchKernelGGL( k_stretch_map_const<-1>, grid, block, 0, st, d_map, F, bins, sr, float( hop ), d_max, marks ); break;
		}
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

extern "C" int flanhip_debug_kernel_scratch_bytes( int which )
	{
	if( int rc = require_device() ) return rc;
	hipFuncAttributes a{};
	const void * fn = which == 0 ? reinterpret_cast<const void*>( k_stretch_map<true> ) : which == 1 ? reinterpret_cast<const void*>( k_stretch_map<false> ) : nullptr;
	FLANHIP_REQUIRE( fn, FLANHIP_ERR_INVALID_ARG, "unknown kernel" );
	FLANHIP_CHECK( hipFuncGetAttributes( &a, fn ) );
	return int( a.localSizeBytes );
	}

static int modify_frequency_dev_impl( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, const float * d_mod,
	const float * d_in_modified, flanhip_MF * d_out, int interp, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( valid_interp( interp ), FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( d_mod && d_in_modified, FLANHIP_ERR_INVALID_ARG, "null map" );
	hipStream_t s = (hipStream_t) stream;
	const int64_t rows = ch * F;
	const size_t per_wave = sizeof( MFd ) * size_t( bins );
	FLANHIP_REQUIRE( per_wave <= 65536, FLANHIP_ERR_UNSUPPORTED, "more than 8192 bins" );
	const int waves = int( std::min<size_t>( 4, 65536 / per_wave ) );
	hipLaunchKernelGGL( ( interp == FLANHIP_INTERP_LINEAR ? k_modify_frequency<false, true> : k_modify_frequency<false, false> ), dim3( (unsigned) ( ( rows + waves - 1 ) / waves ) ), dim3( 64 * waves ), per_wave * waves, s,
		(const MFd*) d_pv, int( ch ), F, bins, sr, float( ( bins - 1 ) * 2 ), d_mod, d_in_modified, (MFd*) d_out, interp );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_modify_frequency_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, const float * d_mod,
	const float * d_in_modified, flanhip_MF * d_out, void * stream )
	{
	return modify_frequency_dev_impl( d_pv, ch, F, bins, sr, d_mod, d_in_modified, d_out, FLANHIP_INTERP_LINEAR, stream );
	}

int flanhip_modify_frequency_interp_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, const float * d_mod,
	const float * d_in_modified, int interp, flanhip_MF * d_out, void * stream )
	{
	return modify_frequency_dev_impl( d_pv, ch, F, bins, sr, d_mod, d_in_modified, d_out, interp, stream );
	}

static int repitch_dev_impl( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float * d_factor, flanhip_MF * d_out, int interp, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( valid_interp( interp ), FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( d_factor, FLANHIP_ERR_INVALID_ARG, "null factor grid" );
	hipStream_t s = (hipStream_t) stream;
	const float dft = float( ( bins - 1 ) * 2 );
	if( int rc = launch_repitch_scan( d_factor, F, bins, sr, dft, s ) ) return rc;                 // PVModify.cpp:278-284
	const int64_t rows = ch * F;
	const size_t per_wave = ( sizeof( MFd ) + sizeof( float ) ) * size_t( bins );
	FLANHIP_REQUIRE( per_wave <= 65536, FLANHIP_ERR_UNSUPPORTED, "more than 5461 bins" );
	const int waves = int( std::min<size_t>( 4, 65536 / per_wave ) );
	hipLaunchKernelGGL( ( interp == FLANHIP_INTERP_LINEAR ? k_modify_frequency<true, true> : k_modify_frequency<true, false> ), dim3( (unsigned) ( ( rows + waves - 1 ) / waves ) ), dim3( 64 * waves ), per_wave * waves, s,
		(const MFd*) d_pv, int( ch ), F, bins, sr, dft, (const float*) d_factor, (const float*) nullptr, (MFd*) d_out, interp );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_repitch_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float * d_factor, flanhip_MF * d_out, void * stream )
	{
	return repitch_dev_impl( d_pv, ch, F, bins, sr, d_factor, d_out, FLANHIP_INTERP_LINEAR, stream );
	}

int flanhip_repitch_interp_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float * d_factor, int interp, flanhip_MF * d_out, void * stream )
	{
	return repitch_dev_impl( d_pv, ch, F, bins, sr, d_factor, d_out, interp, stream );
	}

int flanhip_modify_frequency( const flanhip_MF * pv, int64_t ch, int64_t F, int bins, float sr, const float * mod,
	const float * in_modified, flanhip_MF * out, volatile int * cancel )
	{
	if( int rc = check_pv_args( pv, out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( mod && in_modified, FLANHIP_ERR_INVALID_ARG, "null map" );
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	DevBuf d_pv, d_mod, d_inm, d_out;
	const size_t pv_bytes = sizeof( flanhip_MF ) * size_t( ch ) * F * bins, mod_bytes = sizeof( float ) * size_t( F ) * bins;
	const size_t inm_bytes = sizeof( float ) * size_t( ch ) * F * bins;
	if( int rc = d_pv.alloc( pv_bytes ) ) return rc;
	if( int rc = d_mod.alloc( mod_bytes ) ) return rc;
	if( int rc = d_inm.alloc( inm_bytes ) ) return rc;
	if( int rc = d_out.alloc( pv_bytes ) ) return rc;
	if( int rc_t = flanhip_upload( d_pv.p, pv, pv_bytes ) ) return rc_t;
	if( int rc_t = flanhip_upload( d_mod.p, mod, mod_bytes ) ) return rc_t;
	if( int rc_t = flanhip_upload( d_inm.p, in_modified, inm_bytes ) ) return rc_t;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	if( int rc = flanhip_modify_frequency_dev( (const flanhip_MF*) d_pv.p, ch, F, bins, sr, (const float*) d_mod.p, (const float*) d_inm.p, (flanhip_MF*) d_out.p, nullptr ) ) return rc;
	FLANHIP_CHECK( hipDeviceSynchronize() );
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	if( int rc_t = flanhip_download( out, d_out.p, pv_bytes ) ) return rc_t;
	return FLANHIP_OK;
	}

int flanhip_repitch_map_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float * d_factor,
	float * d_in_modified, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_in_modified, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( d_factor, FLANHIP_ERR_INVALID_ARG, "null factor grid" );
	hipStream_t s = (hipStream_t) stream;
	const float dft = float( ( bins - 1 ) * 2 );
	if( int rc = launch_repitch_scan( d_factor, F, bins, sr, dft, s ) ) return rc;
	const int64_t count = ch * F * bins;
	hipLaunchKernelGGL( k_repitch_lerp, dim3( (unsigned) ( ( count + 255 ) / 256 ) ), dim3( 256 ), 0, s,
		(const MFd*) d_pv, count, F, bins, sr, dft, d_factor, d_in_modified );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

static int shape_common( const flanhip_MF * d_pv, const flanhip_MF * d_tbl, bool affine, int64_t ch, int64_t F, int bins, float sr,
	float a, float b, float c, float d, int align, flanhip_MF * d_out, hipStream_t s )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	const int64_t count = ch * F * bins;
	const float dft = float( ( bins - 1 ) * 2 );
	if( !align )
		{
		const unsigned blocks = (unsigned) ( ( count + 255 ) / 256 );
		if( affine ) hipLaunchKernelGGL( k_shape_plain<true>, dim3( blocks ), dim3( 256 ), 0, s, (const MFd*) d_pv, (const MFd*) nullptr, count, a, b, c, d, (MFd*) d_out );
		else         hipLaunchKernelGGL( k_shape_plain<false>, dim3( blocks ), dim3( 256 ), 0, s, (const MFd*) d_pv, (const MFd*) d_tbl, count, a, b, c, d, (MFd*) d_out );
		}
	else
		{
		const int64_t rows = ch * F;
		const size_t per_wave = sizeof( unsigned long long ) * size_t( bins );
		FLANHIP_REQUIRE( per_wave <= 65536, FLANHIP_ERR_UNSUPPORTED, "shift alignment: more than 8192 bins" );
		const int waves = int( std::min<size_t>( 4, 65536 / per_wave ) );
		const unsigned blocks = (unsigned) ( ( rows + waves - 1 ) / waves );
		if( affine ) hipLaunchKernelGGL( k_shape_aligned<true>, dim3( blocks ), dim3( 64 * waves ), per_wave * waves, s, (const MFd*) d_pv, (const MFd*) nullptr, rows, bins, sr, dft, a, b, c, d, (MFd*) d_out );
		else         hipLaunchKernelGGL( k_shape_aligned<false>, dim3( blocks ), dim3( 64 * waves ), per_wave * waves, s, (const MFd*) d_pv, (const MFd*) d_tbl, rows, bins, sr, dft, a, b, c, d, (MFd*) d_out );
		}
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_shape_affine_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float a, float b, float c, float d,
	int align, flanhip_MF * d_out, void * stream )
	{
	return shape_common( d_pv, nullptr, true, ch, F, bins, sr, a, b, c, d, align, d_out, (hipStream_t) stream );
	}

int flanhip_shape_table_dev( const flanhip_MF * d_pv, const flanhip_MF * d_shaped, int64_t ch, int64_t F, int bins, float sr,
	int align, flanhip_MF * d_out, void * stream )
	{
	FLANHIP_REQUIRE( d_shaped, FLANHIP_ERR_INVALID_ARG, "null shaped table" );
	return shape_common( d_pv, d_shaped, false, ch, F, bins, sr, 0, 0, 0, 0, align, d_out, (hipStream_t) stream );
	}

static int shape_fused( const flanhip_MF * d_pv, const flanhip_MF * d_tbl, bool affine, int64_t ch, int64_t F, int bins, float sr, float analysis_rate,
	float a, float b, float c, float d, flanhip_MF * d_out, int window_size, void * d_ws, hipStream_t s )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( d_ws && analysis_rate > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad workspace / analysis rate" );
	SynthLayout lay;                                                               // the chains convert_to_audio will cut the result into
	if( int rc = synth_layout( ch, F, bins, sr, analysis_rate, window_size, &lay ) ) return rc;
	ShapeChainParams p{};
	p.in = (const MFd*) d_pv; p.tbl = (const MFd*) d_tbl; p.out = (MFd*) d_out;
	p.F = F; p.num_channels = int( ch ); p.bins = bins; p.L = lay.L; p.chains_per_channel = lay.chains_per_channel;
	p.a = a; p.b = b; p.c = c; p.d = d; p.analysis_rate = analysis_rate;
	p.sums = reinterpret_cast<double*>( d_ws );
	p.words = reinterpret_cast<int*>( reinterpret_cast<char*>( d_ws ) + lay.carry_bytes + lay.head_bytes );
	p.epoch = next_epoch();
	note_workspace_producer( d_ws, 0 );                                               // chain sums, no group totals
	const int64_t owners = ch * int64_t( lay.chains_per_channel ) * bins;
	FLANHIP_REQUIRE( ( owners + 255 ) / 256 < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains" );
	if( affine ) hipLaunchKernelGGL( k_shape_plain_chains<true>, dim3( (unsigned) ( ( owners + 255 ) / 256 ) ), dim3( 256 ), 0, s, p );
	else         hipLaunchKernelGGL( k_shape_plain_chains<false>, dim3( (unsigned) ( ( owners + 255 ) / 256 ) ), dim3( 256 ), 0, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_shape_affine_dev_fused( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float analysis_rate, float a, float b, float c, float d,
	flanhip_MF * d_out, int window_size, void * d_ws, void * stream )
	{
	return shape_fused( d_pv, nullptr, true, ch, F, bins, sr, analysis_rate, a, b, c, d, d_out, window_size, d_ws, (hipStream_t) stream );
	}

int flanhip_shape_table_dev_fused( const flanhip_MF * d_pv, const flanhip_MF * d_shaped, int64_t ch, int64_t F, int bins, float sr, float analysis_rate,
	flanhip_MF * d_out, int window_size, void * d_ws, void * stream )
	{
	FLANHIP_REQUIRE( d_shaped, FLANHIP_ERR_INVALID_ARG, "null shaped table" );
	return shape_fused( d_pv, d_shaped, false, ch, F, bins, sr, analysis_rate, 0, 0, 0, 0, d_out, window_size, d_ws, (hipStream_t) stream );
	}

int flanhip_shape_affine( const flanhip_MF * pv, int64_t ch, int64_t F, int bins, float sr, float a, float b, float c, float d,
	int align, flanhip_MF * out, volatile int * cancel )
	{
	if( int rc = check_pv_args( pv, out, ch, F, bins, sr ) ) return rc;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	DevBuf d_pv, d_out;
	const size_t bytes = sizeof( flanhip_MF ) * size_t( ch ) * F * bins;
	if( int rc = d_pv.alloc( bytes ) ) return rc;
	if( int rc = d_out.alloc( bytes ) ) return rc;
	if( int rc_t = flanhip_upload( d_pv.p, pv, bytes ) ) return rc_t;
	if( int rc = flanhip_shape_affine_dev( (const flanhip_MF*) d_pv.p, ch, F, bins, sr, a, b, c, d, align, (flanhip_MF*) d_out.p, nullptr ) ) return rc;
	FLANHIP_CHECK( hipDeviceSynchronize() );
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	if( int rc_t = flanhip_download( out, d_out.p, bytes ) ) return rc_t;
	return FLANHIP_OK;
	}

} // extern "C"
