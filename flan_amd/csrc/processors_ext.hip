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
//   k_desample_select   : same skeleton for the selection accumulator (channel independent), forward then backward over frames:
//                         per (frame, bin) the selected frames on either side; k_desample_apply is then elementwise
//   k_n_loudest         : wavefront per (channel, frame) row: the n-th largest |m| by a 31-step bisection on the bit pattern
//                         (ballot + popcount, keys in LDS), ties by ascending bin
//   k_time_extrapolate  : wavefront per output row, placement conflicts through LDS keys (processors_common.h)
#include "processors_common.h"
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
	__shared__ float lds[3 * kResTF * ( kResTB + 1 )];
	const int bin = blockIdx.x * kResTB + threadIdx.x % kResTB;
	const bool valid = bin < bins;
	const MFd * ip = in + int64_t( blockIdx.y ) * F * bins + bin;
	MFd * op = out + int64_t( blockIdx.y ) * Fo * bins + bin;
	float sm = 0.0f, sf = 0.0f;                                                       // out[frame-1] of this column
	column_scan<kResTF, kResTB, 3, 2, false>( lds, Fo,
		[&]( int64_t f, float ( &v )[3] )
			{
			MFd mf = { __builtin_nanf( "" ), 0.0f };                                  // no input frame: `m > decayed` is false (:633)
			if( valid && f < F ) mf = ip[f * bins];
			v[0] = mf.m; v[1] = mf.f;
			v[2] = ( valid && decay_t ) ? decay_t[f * bins + bin] : decay_t_const;
			},
		[&]( int64_t f, float ( &v )[3] )
			{
			const float decayed_amp = sm * v[2];                                      // :632
			const bool take = ( f == 0 ) || v[0] > decayed_amp;                       // :619-624 (frame 0 is copied), :633
			sm = take ? v[0] : decayed_amp;                                           // :634 / :636
			sf = take ? v[1] : sf;
			v[0] = sm; v[1] = sf;
			},
		[&]( int64_t f, float ( &v )[2] ) { if( valid ) op[f * bins] = MFd{ v[0], v[1] }; } );
	}

// ---------------------------------------------------------------------------------------------------------------------
// retain / remove n loudest partials
// ---------------------------------------------------------------------------------------------------------------------
// predicateNLoudestPartials, PV.cpp:552-590.  One wavefront per (channel, frame) row; s_keys: dynamic LDS, bins u32 per wave.
__global__ __launch_bounds__( 256 ) void k_n_loudest( const MFd * in, int64_t rows, int64_t F, int bins, const int * n_per_frame, int n_const,
	int remove, MFd * out )
	{
	extern __shared__ unsigned s_keys32[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t idx = int64_t( blockIdx.x ) * ( blockDim.x >> 6 ) + wave;
	if( idx >= rows ) return;
	unsigned * keys = s_keys32 + size_t( wave ) * bins;
	const MFd * row = in + idx * bins;
	MFd * orow = out + idx * bins;
	const int64_t frame = idx % F;
	const int64_t n = min( max( int64_t( n_per_frame ? n_per_frame[frame] : n_const ), int64_t( 0 ) ), F );   // :556 (sic: frames)
	for( int bin = lane; bin < bins; bin += 64 ) keys[bin] = __float_as_uint( row[bin].m ) & 0x7fffffffu;      // abs, :571
	wave_sync();

	// T = the n-th largest key: the greatest T with #{ key >= T } >= n, found bit by bit
	unsigned T = 0;
	if( n > 0 && n < bins )
		{
		for( int bit = 30; bit >= 0; --bit )
			{
			const unsigned cand = T | ( 1u << bit );
			int cnt = 0;
			for( int base = 0; base < bins; base += 64 )
				{
				const int bin = base + lane;
				cnt += __popcll( __ballot( bin < bins && keys[min( bin, bins - 1 )] >= cand ) );
				}
			if( cnt >= n ) T = cand;
			}
		}
	int greater = 0;
	for( int base = 0; base < bins; base += 64 )
		{
		const int bin = base + lane;
		greater += __popcll( __ballot( bin < bins && keys[min( bin, bins - 1 )] > T ) );
		}
	const int64_t ties_in = n - greater;                                              // bins equal to T that are among the n loudest
	int seen = 0;                                                                     // equal bins below this 64-bin group
	for( int base = 0; base < bins; base += 64 )
		{
		const int bin = base + lane;
		const bool valid = bin < bins;
		const unsigned key = keys[min( bin, bins - 1 )];
		const bool eq = valid && key == T;
		const unsigned long long mask = __ballot( eq );
		const int rank_eq = seen + __popcll( mask & ( ( 1ull << lane ) - 1ull ) );
		seen += __popcll( mask );
		bool among = n >= bins || ( n > 0 && ( key > T || ( eq && rank_eq < ties_in ) ) );   // rank < n
		const bool keep = remove ? !among : among;                                    // :595 / :601
		if( valid )
			{
			const MFd v = row[bin];
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
	__shared__ float lds[kSelTF * ( kSelTB + 1 )];
	const int bin = blockIdx.x * kSelTB + threadIdx.x % kSelTB;
	const bool valid = bin < bins;
	float accum = 1.0f;                                                               // :461
	int last = -1;
	column_scan<kSelTF, kSelTB, 1, 1, false>( lds, F,
		[&]( int64_t f, float ( &v )[1] ) { v[0] = ( valid && ratio ) ? ratio[f * bins + bin] : ratio_const; },
		[&]( int64_t f, float ( &v )[1] )
			{
			accum += clamp01( v[0] );                                                 // :468-469
			const bool sel = accum >= 1.0f && f < F;                                  // :470
			last = sel ? int( f ) : last;
			accum = sel ? accum - 1.0f : accum;                                       // :473
			v[0] = __int_as_float( last );
			},
		[&]( int64_t f, float ( &v )[1] ) { if( valid ) L[f * bins + bin] = __float_as_int( v[0] ); } );
	}

// backward: R[frame][bin] = the first selected frame > frame (-1: none).  Frame t is selected iff L[t] == t.
__global__ __launch_bounds__( 256 ) void k_desample_next( const int * L, int64_t F, int bins, int * R )
	{
	__shared__ float lds[kSelTF * ( kSelTB + 1 )];
	const int bin = blockIdx.x * kSelTB + threadIdx.x % kSelTB;
	const bool valid = bin < bins;
	int next = -1;
	column_scan<kSelTF, kSelTB, 1, 1, true>( lds, F,
		[&]( int64_t f, float ( &v )[1] ) { v[0] = __int_as_float( valid ? L[f * bins + bin] : -2 ); },
		[&]( int64_t f, float ( &v )[1] )
			{
			const int l = __float_as_int( v[0] );
			v[0] = __int_as_float( next );
			next = ( f < F && l == int( f ) ) ? l : next;
			},
		[&]( int64_t f, float ( &v )[1] ) { if( valid ) R[f * bins + bin] = __float_as_int( v[0] ); } );
	}

// PVModify.cpp:483-506, one thread per output MF
__global__ __launch_bounds__( 256 ) void k_desample_apply( const MFd * in, int64_t F, int bins, int64_t count, const int * L, const int * R, int interp, MFd * out )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= count ) return;
	const int bin = int( idx % bins );
	const int64_t frame = ( idx / bins ) % F;
	const int64_t channel = idx / ( int64_t( bins ) * F );
	const int lFrame = L[frame * bins + bin], rFrame = R[frame * bins + bin];
	if( lFrame < 0 || rFrame < 0 ) { out[idx] = MFd{ 0.0f, 0.0f }; return; }          // :453 clear_buffer, :482
	const MFd lMF = in[( channel * F + lFrame ) * bins + bin];
	const MFd rMF = in[( channel * F + rFrame ) * bins + bin];
	const float mix = interpolate( interp, float( int( frame ) - lFrame ) / float( rFrame - lFrame ) );   // :491
	const float w0 = ( 1.0f - mix ) * lMF.m;
	const float w1 = mix * rMF.m;
	out[idx] = MFd{ w0 + w1, w0 > w1 ? lMF.f : rMF.f };                               // :494-498
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

using namespace flanhip;

extern "C" {

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
	hipLaunchKernelGGL( k_resonate, dim3( ( bins + kResTB - 1 ) / kResTB, (unsigned) ch ), dim3( 256 ), 0, s, (const MFd*) d_pv, F, bins, Fo,
		(const float*) d_decay_t, decay_t_const, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	if( d_decay_t ) FLANHIP_CHECK( hipFreeAsync( d_decay_t, s ) );
	return FLANHIP_OK;
	}

int flanhip_n_loudest_partials_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const int32_t * d_n, int32_t n_const, int remove,
	flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	const size_t per_wave = sizeof( unsigned ) * size_t( bins );
	FLANHIP_REQUIRE( per_wave <= 65536, FLANHIP_ERR_UNSUPPORTED, "more than 16384 bins" );
	const int waves = int( std::min<size_t>( 4, 65536 / per_wave ) );
	const int64_t rows = ch * F;
	hipLaunchKernelGGL( k_n_loudest, dim3( blocks_for( rows, waves ) ), dim3( 64 * waves ), per_wave * waves, (hipStream_t) stream,
		(const MFd*) d_pv, rows, F, bins, (const int*) d_n, int( n_const ), remove ? 1 : 0, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_desample_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const float * d_ratio, float ratio_const, int interp,
	flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	FLANHIP_REQUIRE( interp >= FLANHIP_INTERP_LINEAR && interp <= FLANHIP_INTERP_SINE, FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( F < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "2^31 frames or more" );
	hipStream_t s = (hipStream_t) stream;
	int * d_lr = nullptr;                                                             // L then R: int[2][F][bins]
	const size_t grid = size_t( F ) * bins;
	FLANHIP_CHECK( hipMallocAsync( reinterpret_cast<void**>( &d_lr ), sizeof( int ) * 2 * grid, s ) );
	const unsigned col_blocks = ( bins + kSelTB - 1 ) / kSelTB;
	hipLaunchKernelGGL( k_desample_select, dim3( col_blocks ), dim3( 256 ), 0, s, d_ratio, ratio_const, F, bins, d_lr );
	hipLaunchKernelGGL( k_desample_next, dim3( col_blocks ), dim3( 256 ), 0, s, (const int*) d_lr, F, bins, d_lr + grid );
	const int64_t count = ch * F * bins;
	hipLaunchKernelGGL( k_desample_apply, dim3( blocks_for( count, 256 ) ), dim3( 256 ), 0, s, (const MFd*) d_pv, F, bins, count,
		(const int*) d_lr, (const int*) ( d_lr + grid ), interp, (MFd*) d_out );
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
