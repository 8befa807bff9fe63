// processors_ext.hip -- the remaining embarrassingly parallel PV frame processors behind the C ABI:
// replace_amplitudes, subtract_amplitudes (PV/PV.cpp:205-264), resonate (:604-641), retain/remove_n_loudest_partials
// (:552-602), desample (PV/PVModify.cpp:445-511), time_extrapolate (:607-666).
//
// All of them stream MF[ch][F][bins] once (8 B per MF in, 8 B out): HBM-bound work.  Where the reference defines a result
// through a sequential walk (a decaying running maximum over frames, a selection accumulator over frames, a "strictly louder
// replaces" rule over bins, a sort) the kernel keeps that order along the dependent axis only:
//   k_resonate          : block = 16 bins of one channel; (frame x bin) tiles go through LDS so that all 256 threads move memory
//                         (the next tile is in flight during the scan) while 16 threads carry the recurrence down the tile in
//                         registers (column_scan, processors_common.h)
//   k_desample_select   : same skeleton for the selection accumulator (channel independent): per (frame, bin) the last selected
//                         frame; k_desample_apply finds the next one by bisection and interpolates every channel
//   k_n_loudest         : wavefront per (channel, frame) row: the n-th largest |m| by a 31-step bisection on the bit pattern
//                         (ballot + popcount, keys in LDS), ties by ascending bin
//   k_time_extrapolate  : wavefront per output row, placement conflicts through LDS keys (processors_common.h)
#include "processors_common.h"
#include <mutex>
#include <vector>
#include <algorithm>
#include <algorithm>

namespace flanhip {

// ---------------------------------------------------------------------------------------------------------------------
// replace_amplitudes / subtract_amplitudes
// ---------------------------------------------------------------------------------------------------------------------
template<bool SUBTRACT>
__global__ __launch_bounds__( 256 ) void k_combine_amplitudes( const MFd * in, int64_t F, int bins, int64_t count,
	const MFd * src, int sch, int64_t sF, int sbins, const float * amount, float amount_const, MFd * out )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= count ) return;
	const int bin = int( idx % bins );
	const int64_t frame = ( idx / bins ) % F;
	const int64_t channel = idx / ( int64_t( bins ) * F );
	const MFd cur = in[idx];
	const bool overlap = channel < sch && frame < sF && bin < sbins;                  // PV.cpp:217-219 / :248-250
	if( !overlap ) { out[idx] = SUBTRACT ? cur : MFd{ 0.0f, 0.0f }; return; }        // :246 copy / :215 clear_buffer
	const float a = amount ? amount[frame * bins + bin] : amount_const;
	const float sm = src[( channel * sF + frame ) * sbins + bin].m;
	if( SUBTRACT ) out[idx] = MFd{ fabsf( cur.m - sm * a ), cur.f };                 // :258
	else
		{
		const float amount_c = clamp01( a );                                          // :212
		out[idx] = MFd{ sm * amount_c + cur.m * ( 1.0f - amount_c ), cur.f };         // :230-233
		}
	}

// ---------------------------------------------------------------------------------------------------------------------
// resonate
// ---------------------------------------------------------------------------------------------------------------------
// PV.cpp:617,:631 for a sampled decay grid: decay_t = pow( clamp( decay, 0, 1 ), seconds per frame ), correctly rounded
// (evaluated in double, rounded once).
__global__ __launch_bounds__( 256 ) void k_decay_pow( const float * decay, int64_t count, float seconds_per_frame, float * decay_t )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= count ) return;
	decay_t[idx] = float( pow( double( clamp01( decay[idx] ) ), double( seconds_per_frame ) ) );
	}

constexpr int kResTB = 16, kResTF = 128;                                            // k_resonate tile: 16 bins x 128 frames (3 x 8.5 KB of LDS)
// PV.cpp:619-638.  grid = ( ceil(bins/16), channels ).  decay_t: float[Fo][bins] or nullptr + constant.  column_scan over
// the OUTPUT's frames with ( m, f, decay_t ) in and ( m, f ) out.
__global__ __launch_bounds__( 256 ) void k_resonate( const MFd * in, int64_t F, int bins, int64_t Fo, const float * decay_t, float decay_t_const, MFd * out )
	{
	__shared__ __attribute__(( aligned( 16 ) )) float lds[column_scan_lds_floats( kResTF, kResTB, 3 )];
	const int strip = xcd_contiguous_strip( blockIdx.x, ( bins + kResTB - 1 ) / kResTB );
	if( strip < 0 ) return;
	const int bin = strip * kResTB + threadIdx.x % kResTB;
	const bool valid = bin < bins;
	const MFd * ip = in + int64_t( blockIdx.y ) * F * bins + bin;
	MFd * op = out + int64_t( blockIdx.y ) * Fo * bins + bin;
	// out[frame-1] of this column.  Frame 0 is a copy of the input (:619-624): the state starts AS frame 0 and that step
	// multiplies by 1, which leaves it in place -- the recurrence then has no special case inside its dependent chain.
	float sm = 0.0f, sf = 0.0f;
	if( valid && threadIdx.x < kResTB ) { const MFd first = ip[0]; sm = first.m; sf = first.f; }
	column_scan<kResTF, kResTB, 3, 2, false>( lds, Fo,
		[&]( int64_t f, float ( &v )[3] )
			{
			MFd mf = { __builtin_nanf( "" ), 0.0f };                                  // no input frame: `m > decayed` is false (:633)
			if( valid && f < F ) mf = ip[f * bins];
			v[0] = mf.m; v[1] = mf.f;
			v[2] = ( valid && decay_t ) ? decay_t[f * bins + bin] : decay_t_const;
			if( f == 0 ) { v[0] = __builtin_nanf( "" ); v[2] = 1.0f; }
			},
		[&]( int64_t f, float ( &v )[3] )
			{
			const float decayed_amp = sm * v[2];                                      // :632
			const bool take = v[0] > decayed_amp;                                     // :633
			sm = take ? v[0] : decayed_amp;                                           // :634 / :636
			sf = take ? v[1] : sf;
			v[0] = sm; v[1] = sf;
			},
		[&]( int64_t f, float ( &v )[2] ) { if( valid ) op[f * bins] = MFd{ v[0], v[1] }; } );
	}

// ---------------------------------------------------------------------------------------------------------------------
// retain / remove n loudest partials
// ---------------------------------------------------------------------------------------------------------------------
// predicateNLoudestPartials, PV.cpp:552-590.  One wavefront per (channel, frame) row.  The row's keys ( |m| bit patterns )
// live in registers when bins <= 64 Q (Q = 3 ... 33: dft 256 ... 4096), else in LDS (s_keys32: bins u32 per wave).
template<int Q>
__global__ __launch_bounds__( 256 ) void k_n_loudest( const MFd * in, int64_t rows, int64_t F, int bins, const int * n_per_frame, int n_const,
	int remove, MFd * out )
	{
	extern __shared__ unsigned s_keys32[];
	constexpr bool REG = Q > 0;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t idx = int64_t( blockIdx.x ) * ( blockDim.x >> 6 ) + wave;
	if( idx >= rows ) return;
	const MFd * row = in + idx * bins;
	MFd * orow = out + idx * bins;
	const int64_t frame = idx % F;
	const int64_t n = min( max( int64_t( n_per_frame ? n_per_frame[frame] : n_const ), int64_t( 0 ) ), F );   // :556 (sic: frames)
	const int groups = REG ? Q : ( bins + 63 ) / 64;
	unsigned rkey[REG ? Q : 1];
	MFd rmf[REG ? Q : 1];
	unsigned * keys = s_keys32 + ( REG ? 0 : size_t( wave ) * bins );
	if constexpr( REG )
		{
		#pragma unroll
		for( int q = 0; q < Q; ++q )
			{
			const int bin = lane + 64 * q;
			rmf[q] = bin < bins ? row[bin] : MFd{ 0.0f, 0.0f };
			rkey[q] = __float_as_uint( rmf[q].m ) & 0x7fffffffu;                      // abs, :571
			}
		}
	else
		{
		for( int bin = lane; bin < bins; bin += 64 ) keys[bin] = __float_as_uint( row[bin].m ) & 0x7fffffffu;
		wave_sync();
		}
	auto key_of = [&]( int q ) { return REG ? rkey[REG ? q : 0] : keys[min( lane + 64 * q, bins - 1 )]; };
	// number of bins of the row satisfying a predicate on ( key, valid )
	auto count = [&]( auto pred )
		{
		int cnt = 0;
		#pragma unroll
		for( int q = 0; q < groups; ++q ) cnt += __popcll( __ballot( lane + 64 * q < bins && pred( key_of( q ) ) ) );
		return cnt;
		};

	// T = the n-th largest key: the greatest T with #{ key >= T } >= n, found bit by bit
	unsigned T = 0;
	if( n > 0 && n < bins )
		for( int bit = 30; bit >= 0; --bit )
			{
			const unsigned cand = T | ( 1u << bit );
			if( count( [&]( unsigned k ) { return k >= cand; } ) >= n ) T = cand;
			}
	const int64_t ties_in = n - count( [&]( unsigned k ) { return k > T; } );         // bins equal to T that are among the n loudest
	int seen = 0;                                                                     // equal bins below this 64-bin group
	#pragma unroll
	for( int q = 0; q < groups; ++q )
		{
		const int bin = lane + 64 * q;
		const bool valid = bin < bins;
		const unsigned key = key_of( q );
		const bool eq = valid && key == T;
		const unsigned long long mask = __ballot( eq );
		const int rank_eq = seen + __popcll( mask & ( ( 1ull << lane ) - 1ull ) );
		seen += __popcll( mask );
		const bool among = n >= bins || ( n > 0 && ( key > T || ( eq && rank_eq < ties_in ) ) );   // rank < n
		const bool keep = remove ? !among : among;                                    // :595 / :601
		if( valid )
			{
			const MFd v = REG ? rmf[REG ? q : 0] : row[bin];
			orow[bin] = keep ? v : MFd{ 0.0f, v.f };                                  // :581-584
			}
		}
	}

// ---------------------------------------------------------------------------------------------------------------------
// desample
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSelTB = 16, kSelTF = 896;                                            // selection scans: 16 bins x 896 frames (61 KB of LDS)
// PVModify.cpp:461-475, forward: L[frame][bin] = the last selected frame <= frame (-1: none yet).  grid = ceil(bins/16).
__global__ __launch_bounds__( 256 ) void k_desample_select( const float * ratio, float ratio_const, int64_t F, int bins, int * L )
	{
	__shared__ __attribute__(( aligned( 16 ) )) float lds[column_scan_lds_floats( kSelTF, kSelTB, 1 )];
	const int strip = xcd_contiguous_strip( blockIdx.x, ( bins + kSelTB - 1 ) / kSelTB );
	if( strip < 0 ) return;
	const int bin = strip * kSelTB + threadIdx.x % kSelTB;
	const bool valid = bin < bins;
	float accum = 1.0f;                                                               // :461
	int last = -1;
	column_scan<kSelTF, kSelTB, 1, 1, false>( lds, F,
		[&]( int64_t f, float ( &v )[1] ) { v[0] = clamp01( ( valid && ratio ) ? ratio[f * bins + bin] : ratio_const ); },   // :468
		[&]( int64_t f, float ( &v )[1] )
			{
			accum += v[0];                                                            // :469
			const bool sel = accum >= 1.0f;                                           // :470 (steps past the last frame are never stored)
			last = sel ? int( f ) : last;
			accum = sel ? accum - 1.0f : accum;                                       // :473
			v[0] = __int_as_float( last );
			},
		[&]( int64_t f, float ( &v )[1] ) { if( valid ) L[f * bins + bin] = __float_as_int( v[0] ); } );
	}

// PVModify.cpp:483-506, one thread per (frame, bin): the endpoints and the mix are the same for every channel
// The right endpoint is the first frame after `frame` whose L exceeds L[frame] (L is non-decreasing down a column and jumps to t
// exactly at a selected frame t): found by bisection, 13 probes of an L2-resident grid -- cheaper than a second sequential scan.
__global__ __launch_bounds__( 256 ) void k_desample_apply( const MFd * in, int num_channels, int64_t F, int bins, const int * L, int interp, MFd * out )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	const int64_t per_channel = F * bins;
	if( idx >= per_channel ) return;
	const int bin = int( idx % bins );
	const int64_t frame = idx / bins;
	const int before = L[idx];
	int after = -1;
	if( before >= 0 )
		{
		int64_t lo = frame + 1, hi = F;                                             // answer in [frame + 1, F]; F = none
		while( lo < hi )
			{
			const int64_t mid = ( lo + hi ) >> 1;
			if( L[mid * bins + bin] > before ) hi = mid; else lo = mid + 1;
			}
		if( lo < F ) after = int( lo );
		}
	const bool none = before < 0 || after < 0;                                       // :453 clear_buffer, :482
	const float mix = none ? 0.0f : interpolate( interp, float( int( frame ) - before ) / float( after - before ) );   // :491
	for( int channel = 0; channel < num_channels; ++channel )
		{
		MFd o = { 0.0f, 0.0f };
		if( !none )
			{
			const MFd a = in[( int64_t( channel ) * F + before ) * bins + bin];
			const MFd b = in[( int64_t( channel ) * F + after ) * bins + bin];
			const float share_a = ( 1.0f - mix ) * a.m;
			const float share_b = mix * b.m;
			o = MFd{ share_a + share_b, share_a > share_b ? a.f : b.f };                              // :494-498
			}
		out[int64_t( channel ) * per_channel + idx] = o;
		}
	}

// ---------------------------------------------------------------------------------------------------------------------
// time_extrapolate
// ---------------------------------------------------------------------------------------------------------------------
// PVModify.cpp:635-664.  One wavefront per (channel, output frame) row; s_keys: dynamic LDS, bins u64 per wave.
__global__ __launch_bounds__( 256 ) void k_time_extrapolate( const MFd * in, int64_t F, int bins, float sr, float dft, int64_t start_frame, int64_t end_frame,
	int64_t Fo, int64_t rows, const float * interp_samples, MFd * out )
	{
	extern __shared__ unsigned long long s_keys[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t idx = int64_t( blockIdx.x ) * ( blockDim.x >> 6 ) + wave;
	if( idx >= rows ) return;
	const int64_t channel = idx / Fo, frame = idx % Fo;
	MFd * orow = out + idx * bins;
	if( frame < start_frame )                                                         // :638
		{
		const MFd * row = in + ( channel * F + frame ) * bins;
		for( int bin = lane; bin < bins; bin += 64 ) orow[bin] = row[bin];
		return;
		}
	unsigned long long * keys = s_keys + size_t( wave ) * bins;
	const MFd * lrow = in + ( channel * F + start_frame ) * bins;
	const MFd * rrow = in + ( channel * F + end_frame ) * bins;
	const float mix = interp_samples[frame - start_frame];                            // :643
	auto extrapolated = [&]( int bin )                                                // :656-657
		{
		const MFd leftMF = lrow[bin], rightMF = rrow[bin];
		return MFd{ fabsf( ( 1.0f - mix ) * leftMF.m + mix * rightMF.m ), ( 1.0f - mix ) * leftMF.f + mix * rightMF.f };
		};
	for( int bin = lane; bin < bins; bin += 64 ) keys[bin] = 0ull;                    // :628 clear_buffer
	wave_sync();
	for( int bin = lane; bin < bins; bin += 64 )
		{
		const int rightBinShift = to_int_sat( float( bin ) - frequency_to_bin( rrow[bin].f, sr, dft ) );   // :654
		const MFd e = extrapolated( bin );
		const int extrapBin = to_int_sat( frequency_to_bin( e.f, sr, dft ) + float( rightBinShift ) );     // :658
		if( extrapBin < 0 || extrapBin >= bins ) continue;                            // :659
		placement_offer( keys, extrapBin, e.m, bin );                                 // :663-664
		}
	wave_sync();
	for( int bin = lane; bin < bins; bin += 64 )
		{
		const unsigned long long key = keys[bin];
		orow[bin] = key ? extrapolated( placement_winner( key ) ) : MFd{ 0.0f, 0.0f };
		}
	}

static unsigned blocks_for( int64_t count, int per_block ) { return (unsigned) ( ( count + per_block - 1 ) / per_block ); }

} // namespace flanhip

namespace flanhip {

int processors_ext_set_interp_lut( int slot, const float * d_table ) { return set_interp_lut_here( slot, d_table ); }

// the tables of flanhip_interp_table_create: device copy for the kernels, host copy for the one-value evaluations
struct InterpTable { float * d = nullptr; std::vector<float> host; int device = -1; };
static std::mutex g_lut_mutex;
static InterpTable g_luts[kInterpLutSlots];

bool valid_interp( int kind )
	{
	if( kind >= FLANHIP_INTERP_LINEAR && kind <= FLANHIP_INTERP_SINE ) return true;
	if( kind < FLANHIP_INTERP_TABLE_FIRST || kind >= FLANHIP_INTERP_TABLE_FIRST + kInterpLutSlots ) return false;
	// a table lives on the device it was created on (its pointer is published to that device's copies of g_interp_lut only): on any other
	// device the kind is unknown
	int device = -1;
	if( hipGetDevice( &device ) != hipSuccess ) return false;
	std::lock_guard<std::mutex> lock( g_lut_mutex );
	const InterpTable & t = g_luts[kind - FLANHIP_INTERP_TABLE_FIRST];
	return t.d != nullptr && t.device == device;
	}

float interp_eval_host( int kind, float x )
	{
	if( kind >= FLANHIP_INTERP_TABLE_FIRST )
		{
		std::lock_guard<std::mutex> lock( g_lut_mutex );
		const std::vector<float> & t = g_luts[kind - FLANHIP_INTERP_TABLE_FIRST].host;
		if( t.empty() ) return x;
		if( !( x == x ) ) return t[FLANHIP_INTERP_TABLE_INTERVALS + 1];
		const float pos = ( x < 0.0f ? 0.0f : ( 1.0f < x ? 1.0f : x ) ) * float( FLANHIP_INTERP_TABLE_INTERVALS );
		const int i = std::min( int( pos ), FLANHIP_INTERP_TABLE_INTERVALS - 1 );
		return std::fmaf( pos - float( i ), t[i + 1] - t[i], t[i] );
		}
	switch( kind )                                                                     // Utility/Interpolator.cpp:14-101
		{
		case FLANHIP_INTERP_MIDPOINT: return 0.5f;
		case FLANHIP_INTERP_NEAREST: return std::round( x );
		case FLANHIP_INTERP_FLOOR: return 0.0f;
		case FLANHIP_INTERP_CEIL: return 1.0f;
		case FLANHIP_INTERP_SMOOTHSTEP: return x * x * ( 3.0f - 2.0f * x );
		case FLANHIP_INTERP_SMOOTHERSTEP: return x * x * x * ( x * ( x * 6.0f - 15.0f ) + 10.0f );
		case FLANHIP_INTERP_SQRT: return std::sqrt( x );
		case FLANHIP_INTERP_SINE: return ( 1.0f - std::cos( std::acos( -1.0f ) * x ) ) / 2.0f;
		default: return x;
		}
	}

} // namespace flanhip

using namespace flanhip;

extern "C" {

int flanhip_interp_table_create( const float * samples, int * kind )
	{
	FLANHIP_REQUIRE( samples && kind, FLANHIP_ERR_INVALID_ARG, "null argument" );
	if( int rc = require_device() ) return rc;
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	const size_t count = size_t( FLANHIP_INTERP_TABLE_INTERVALS ) + 2;
	std::lock_guard<std::mutex> lock( g_lut_mutex );
	int slot = -1;
	for( int i = 0; i < kInterpLutSlots && slot < 0; ++i ) if( !g_luts[i].d ) slot = i;
	FLANHIP_REQUIRE( slot >= 0, FLANHIP_ERR_UNSUPPORTED, "all interpolator table slots are in use (flanhip_interp_table_destroy)" );
	float * d = nullptr;
	FLANHIP_CHECK( hipMalloc( &d, sizeof( float ) * count ) );
	int rc = FLANHIP_OK;
	if( hipMemcpy( d, samples, sizeof( float ) * count, hipMemcpyHostToDevice ) != hipSuccess ) { set_error( "hipMemcpy failed" ); rc = FLANHIP_ERR_HIP; }
	if( !rc ) rc = processors_set_interp_lut( slot, d );
	if( !rc ) rc = processors_ext_set_interp_lut( slot, d );
	if( !rc ) rc = processors_arrange_set_interp_lut( slot, d );
	if( rc ) { (void) hipFree( d ); return rc; }
	g_luts[slot].d = d;
	g_luts[slot].host.assign( samples, samples + count );
	g_luts[slot].device = device;
	*kind = FLANHIP_INTERP_TABLE_FIRST + slot;
	return FLANHIP_OK;
	}

int flanhip_interp_table_destroy( int kind )
	{
	FLANHIP_REQUIRE( kind >= FLANHIP_INTERP_TABLE_FIRST && kind < FLANHIP_INTERP_TABLE_FIRST + kInterpLutSlots, FLANHIP_ERR_INVALID_ARG, "not a table kind" );
	std::lock_guard<std::mutex> lock( g_lut_mutex );
	InterpTable & t = g_luts[kind - FLANHIP_INTERP_TABLE_FIRST];
	FLANHIP_REQUIRE( t.d, FLANHIP_ERR_INVALID_ARG, "table not alive" );
	// on the table's OWN device: kernels that read it may still be running there; its entries in that device's g_interp_lut copies are
	// cleared so that a stale kind can never reach freed memory
	int current = -1;
	FLANHIP_CHECK( hipGetDevice( &current ) );
	if( current != t.device ) FLANHIP_CHECK( hipSetDevice( t.device ) );
	const int slot = kind - FLANHIP_INTERP_TABLE_FIRST;
	int rc = FLANHIP_OK;
	if( hipDeviceSynchronize() != hipSuccess ) { set_error( "hipDeviceSynchronize failed" ); rc = FLANHIP_ERR_HIP; }
	if( !rc ) rc = processors_set_interp_lut( slot, nullptr );
	if( !rc ) rc = processors_ext_set_interp_lut( slot, nullptr );
	if( !rc ) rc = processors_arrange_set_interp_lut( slot, nullptr );
	(void) hipFree( t.d );
	t.d = nullptr; t.host.clear(); t.device = -1;
	if( current != -1 ) (void) hipSetDevice( current );
	return rc;
	}


static int combine_amplitudes( bool subtract, const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const flanhip_MF * d_src, int64_t sch, int64_t sF,
	int sbins, const float * d_amount, float amount_const, flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	FLANHIP_REQUIRE( d_src && sch > 0 && sF > 0 && sbins > 0, FLANHIP_ERR_INVALID_ARG, "bad amplitude source" );
	const int64_t count = ch * F * bins;
	hipStream_t s = (hipStream_t) stream;
	if( subtract )
		hipLaunchKernelGGL( k_combine_amplitudes<true>, dim3( blocks_for( count, 256 ) ), dim3( 256 ), 0, s, (const MFd*) d_pv, F, bins, count,
			(const MFd*) d_src, int( sch ), sF, sbins, d_amount, amount_const, (MFd*) d_out );
	else
		hipLaunchKernelGGL( k_combine_amplitudes<false>, dim3( blocks_for( count, 256 ) ), dim3( 256 ), 0, s, (const MFd*) d_pv, F, bins, count,
			(const MFd*) d_src, int( sch ), sF, sbins, d_amount, amount_const, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_replace_amplitudes_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const flanhip_MF * d_src, int64_t sch, int64_t sF, int sbins,
	const float * d_amount, float amount_const, flanhip_MF * d_out, void * stream )
	{
	return combine_amplitudes( false, d_pv, ch, F, bins, d_src, sch, sF, sbins, d_amount, amount_const, d_out, stream );
	}

int flanhip_subtract_amplitudes_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const flanhip_MF * d_src, int64_t sch, int64_t sF, int sbins,
	const float * d_amount, float amount_const, flanhip_MF * d_out, void * stream )
	{
	return combine_amplitudes( true, d_pv, ch, F, bins, d_src, sch, sF, sbins, d_amount, amount_const, d_out, stream );
	}

int64_t flanhip_resonate_out_frames( int64_t F, float length, float sr, int hop )
	{
	if( F <= 0 || hop < 1 || !( sr > 0.0f ) ) return -1;
	if( length < 0 ) length = 0;                                                      // PV.cpp:609-610
	const float extra = std::ceil( length * sr / float( hop ) );                      // time_to_frame, PVBuffer.cpp:428-431
	const float total = float( int32_t( F ) ) + extra;                                // :613
	if( !( total < 2147483648.0f ) ) return -1;
	return int64_t( total );
	}

int flanhip_resonate_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int hop, int64_t Fo, const float * d_decay,
	float decay_const, flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( hop >= 1 && Fo >= F, FLANHIP_ERR_INVALID_ARG, "bad hop / output length" );
	hipStream_t s = (hipStream_t) stream;
	const float seconds_per_frame = 1.0f / ( sr / float( hop ) );                     // frame_to_time( 1 ), PV.cpp:626
	float * d_decay_t = nullptr;
	float decay_t_const = 0.0f;
	if( d_decay )
		{
		const int64_t count = Fo * bins;
		FLANHIP_CHECK( hipMallocAsync( reinterpret_cast<void**>( &d_decay_t ), sizeof( float ) * size_t( count ), s ) );
		hipLaunchKernelGGL( k_decay_pow, dim3( blocks_for( count, 256 ) ), dim3( 256 ), 0, s, d_decay, count, seconds_per_frame, d_decay_t );
		FLANHIP_CHECK( hipGetLastError() );
		}
	else
		{
		const float d = decay_const < 0.0f ? 0.0f : ( 1.0f < decay_const ? 1.0f : decay_const );   // :617
		decay_t_const = std::pow( d, seconds_per_frame );                             // :631, the platform's powf like the reference
		}
	hipLaunchKernelGGL( k_resonate, dim3( xcd_grid( ( bins + kResTB - 1 ) / kResTB ), (unsigned) ch ), dim3( 256 ), 0, s, (const MFd*) d_pv, F, bins, Fo,
		(const float*) d_decay_t, decay_t_const, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	if( d_decay_t ) FLANHIP_CHECK( hipFreeAsync( d_decay_t, s ) );
	return FLANHIP_OK;
	}

int flanhip_n_loudest_partials_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const int32_t * d_n, int32_t n_const, int remove,
	flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	const int64_t rows = ch * F;
	hipStream_t s = (hipStream_t) stream;
	const int q = ( bins + 63 ) / 64;
	#define FLANHIP_NL( Q ) hipLaunchKernelGGL( k_n_loudest<Q>, dim3( blocks_for( rows, 4 ) ), dim3( 256 ), 0, s, (const MFd*) d_pv, rows, F, bins, \
		(const int*) d_n, int( n_const ), remove ? 1 : 0, (MFd*) d_out )
	if( q <= 3 ) FLANHIP_NL( 3 );
	else if( q <= 5 ) FLANHIP_NL( 5 );
	else if( q <= 9 ) FLANHIP_NL( 9 );
	else if( q <= 17 ) FLANHIP_NL( 17 );
	else if( q <= 33 ) FLANHIP_NL( 33 );
	else
		{
		const size_t per_wave = sizeof( unsigned ) * size_t( bins );
		FLANHIP_REQUIRE( per_wave <= 65536, FLANHIP_ERR_UNSUPPORTED, "more than 16384 bins" );
		const int waves = int( std::min<size_t>( 4, 65536 / per_wave ) );
		hipLaunchKernelGGL( k_n_loudest<0>, dim3( blocks_for( rows, waves ) ), dim3( 64 * waves ), per_wave * waves, s,
			(const MFd*) d_pv, rows, F, bins, (const int*) d_n, int( n_const ), remove ? 1 : 0, (MFd*) d_out );
		}
	#undef FLANHIP_NL
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_desample_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const float * d_ratio, float ratio_const, int interp,
	flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	FLANHIP_REQUIRE( valid_interp( interp ), FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( F < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "2^31 frames or more" );
	hipStream_t s = (hipStream_t) stream;
	int * d_lr = nullptr;                                                             // L: int[F][bins]
	const size_t grid = size_t( F ) * bins;
	FLANHIP_CHECK( hipMallocAsync( reinterpret_cast<void**>( &d_lr ), sizeof( int ) * grid, s ) );
	const unsigned col_blocks = xcd_grid( ( bins + kSelTB - 1 ) / kSelTB );
	hipLaunchKernelGGL( k_desample_select, dim3( col_blocks ), dim3( 256 ), 0, s, d_ratio, ratio_const, F, bins, d_lr );
	hipLaunchKernelGGL( k_desample_apply, dim3( blocks_for( int64_t( grid ), 256 ) ), dim3( 256 ), 0, s, (const MFd*) d_pv, int( ch ), F, bins,
		(const int*) d_lr, interp, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	FLANHIP_CHECK( hipFreeAsync( d_lr, s ) );
	return FLANHIP_OK;
	}

int flanhip_time_extrapolate_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int64_t start_frame, int64_t end_frame,
	int64_t Fo, const float * d_interp_samples, flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( d_interp_samples, FLANHIP_ERR_INVALID_ARG, "null interpolator samples" );
	FLANHIP_REQUIRE( 0 <= start_frame && start_frame < end_frame && end_frame < F && Fo > end_frame, FLANHIP_ERR_INVALID_ARG, "bad frame range" );
	const size_t per_wave = sizeof( unsigned long long ) * size_t( bins );
	FLANHIP_REQUIRE( per_wave <= 65536, FLANHIP_ERR_UNSUPPORTED, "more than 8192 bins" );
	const int waves = int( std::min<size_t>( 4, 65536 / per_wave ) );
	const int64_t rows = ch * Fo;
	hipLaunchKernelGGL( k_time_extrapolate, dim3( blocks_for( rows, waves ) ), dim3( 64 * waves ), per_wave * waves, (hipStream_t) stream,
		(const MFd*) d_pv, F, bins, sr, float( ( bins - 1 ) * 2 ), start_frame, end_frame, Fo, rows, d_interp_samples, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

} // extern "C"
