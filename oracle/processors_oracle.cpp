// processors_oracle.cpp -- CPU restatement of the remaining embarrassingly parallel PV frame processors
// (SURVEY 8f rank 4): replace_amplitudes, subtract_amplitudes, resonate, retain/remove_n_loudest_partials,
// desample, time_extrapolate.
//
// TEST INFRASTRUCTURE ONLY (same rules as flan_oracle.cpp: only tests/, smoke() and bench.py's cpu_baseline leg may
// use it).  Every function cites the reference file:line it follows (paths relative to /root/reference/src/flan).
//
// Pinning status: PV/PV.cpp and PV/PVModify.cpp are UNBUILDABLE here (FFTW3f, libsndfile, MSVC-only std::_Pi) and the
// reference ships no tests or vectors for these methods: PARITY UNPINNED by reference fixtures.  The restatement is
// literal (same loops, same fp32 expressions, same order); where the reference is undefined or implementation
// defined the choice made is stated at the function.
//
// User functions (Function<TF,float>, Function<Second,Bin>, Interpolator) are sampled by the CALLER, the way the
// reference samples them on the host before its loops (PV.h:32-35, Function.h:141-171); this file takes the grids.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

namespace {

struct MF { float m, f; };

// PVBuffer.cpp:428-446
float frequency_to_bin( float f, float sr, int dft ) { return f / ( float( sr ) / float( dft ) ); }
float time_to_frame( float t, float sr, int hop ) { return t * float( sr ) / float( hop ); }
float frame_to_time( float f, float sr, int hop ) { return f / ( float( sr ) / float( hop ) ); }

const float k_pi = std::acos( -1.0f );                                  // Utility/Interpolator.cpp:10
const float k_sqrt2 = std::sqrt( 2.0f );                                // :11

// float -> Bin / Frame (int32) conversions of the reference are plain C++ conversions: undefined outside the int range.
// Checker and device both saturate (and map NaN to INT_MIN, which every range check then rejects).
int32_t to_int( float v )
	{
	if( !( v == v ) ) return INT32_MIN;
	if( v >= 2147483648.0f ) return INT32_MAX;
	if( v <= -2147483648.0f ) return INT32_MIN;
	return int32_t( v );
	}

// Utility/Interpolator.cpp:14-110.  kind: 0 linear, 1 midpoint, 2 nearest, 3 floor, 4 ceil, 5 smoothstep, 6 smootherstep,
// 7 sqrt, 8 sine
float interpolate( int kind, float x )
	{
	switch( kind )
		{
		case 0: return x;                                                           // :50-56
		case 1: return 0.5f;                                                        // :14-20
		case 2: return std::round( x );                                             // :23-29
		case 3: return 0.0f;                                                        // :32-38
		case 4: return 1.0f;                                                        // :41-47
		case 5: return x * x * ( 3.0f - 2.0f * x );                                 // :59-65
		case 6: return x * x * x * ( x * ( x * 6.0f - 15.0f ) + 10.0f );            // :68-74
		case 7: return std::sqrt( x );                                              // :95-101
		case 8: return ( 1.0f - std::cos( k_pi * x ) ) / 2.0f;                      // :77-83 (cosf: libm specific)
		// stand-ins for a USER's callable (Interpolator( fn ), Utility/Interpolator.h): the tests hand the same functions to the library
		case 100: return x * x;
		case 101: return x < 0.5f ? 2.0f * x * x : 1.0f - 2.0f * ( 1.0f - x ) * ( 1.0f - x );
		case 102: return x < 0.3f ? 0.0f : 1.0f;
		}
	return x;
	}

} // namespace

extern "C" {

float oracle_interpolate( int kind, float x ) { return interpolate( kind, x ); }

// PV::replace_amplitudes, PV.cpp:205-236.  amount: float[F][bins] sampled over THIS pv's domain (:211), clamped to [0,1]
// here (:212).  out (this pv's format) is cleared (:215); only the overlap with amp_source is written (:217-219).
int oracle_replace_amplitudes( const float * pv_mf, int ch, int64_t F, int bins, const float * src_mf, int sch, int64_t sF, int sbins,
	const float * amount, float * out_mf )
	{
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	const MF * src = reinterpret_cast<const MF*>( src_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::memset( out, 0, sizeof( MF ) * size_t( ch ) * F * bins );
	const int nc = std::min( ch, sch ), nb = std::min( bins, sbins );
	const int64_t nf = std::min( F, sF );
	for( int c = 0; c < nc; ++c )
		for( int64_t frame = 0; frame < nf; ++frame )
			for( int bin = 0; bin < nb; ++bin )
				{
				const MF cur = in[( int64_t( c ) * F + frame ) * bins + bin];
				const float amount_c = std::clamp( amount[frame * bins + bin], 0.0f, 1.0f );
				out[( int64_t( c ) * F + frame ) * bins + bin] =
					{ src[( int64_t( c ) * sF + frame ) * sbins + bin].m * amount_c + cur.m * ( 1.0f - amount_c ), cur.f };   // :230-233
				}
	return 0;
	}

// PV::subtract_amplitudes, PV.cpp:238-264.  out = copy (:246); the amount is NOT clamped (:244).
int oracle_subtract_amplitudes( const float * pv_mf, int ch, int64_t F, int bins, const float * src_mf, int sch, int64_t sF, int sbins,
	const float * amount, float * out_mf )
	{
	const MF * src = reinterpret_cast<const MF*>( src_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::memcpy( out, pv_mf, sizeof( MF ) * size_t( ch ) * F * bins );
	const int nc = std::min( ch, sch ), nb = std::min( bins, sbins );
	const int64_t nf = std::min( F, sF );
	for( int c = 0; c < nc; ++c )
		for( int64_t frame = 0; frame < nf; ++frame )
			for( int bin = 0; bin < nb; ++bin )
				{
				MF & o = out[( int64_t( c ) * F + frame ) * bins + bin];
				const float amount_c = amount[frame * bins + bin];
				o.m = std::abs( o.m - src[( int64_t( c ) * sF + frame ) * sbins + bin].m * amount_c );   // :258
				}
	return 0;
	}

// PV::resonate, PV.cpp:604-641: output frame count, :613 ( int + float -> float -> Frame )
int64_t oracle_resonate_out_frames( int64_t F, float length, float sr, int hop )
	{
	if( length < 0 ) length = 0;                                                    // :609-610
	return int64_t( to_int( float( int32_t( F ) ) + std::ceil( time_to_frame( length, sr, hop ) ) ) );
	}

// decay: float[Fo][bins] sampled over the OUTPUT's domain (:616), clamped to [0,1] here (:617).
// pow_mode 0: std::pow(float,float) = libm powf, what the reference calls (:631; platform specific, <1 ulp);
// pow_mode 1: the correctly rounded power (evaluated in double, rounded once) -- what the device kernel computes for
// a sampled decay grid.  For a constant decay the device library raises the constant on the host with the same
// libm powf, so mode 0 is the comparison there.
int oracle_resonate( const float * pv_mf, int ch, int64_t F, int bins, float sr, int hop, int64_t Fo, const float * decay, int pow_mode,
	float * out_mf )
	{
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	const float secondsPerFrame_c = frame_to_time( 1, sr, hop );                    // :626
	for( int c = 0; c < ch; ++c )
		for( int bin = 0; bin < bins; ++bin )
			{
			MF prev = in[( int64_t( c ) * F + 0 ) * bins + bin];                    // :620-624
			out[( int64_t( c ) * Fo + 0 ) * bins + bin] = prev;
			for( int64_t frame = 1; frame < Fo; ++frame )                           // :630-638
				{
				const float d = std::clamp( decay[frame * bins + bin], 0.0f, 1.0f );
				const float decay_t = pow_mode == 0 ? std::pow( d, secondsPerFrame_c )
				                                    : float( std::pow( double( d ), double( secondsPerFrame_c ) ) );
				const float decayed_amp = prev.m * decay_t;
				MF o;
				if( frame < F && in[( int64_t( c ) * F + frame ) * bins + bin].m > decayed_amp )
					o = in[( int64_t( c ) * F + frame ) * bins + bin];
				else
					o = { decayed_amp, prev.f };
				out[( int64_t( c ) * Fo + frame ) * bins + bin] = o;
				prev = o;
				}
			}
	return 0;
	}

// predicateNLoudestPartials, PV.cpp:552-590 (retain: rank < n, :595; remove: rank >= n, :601).
// n: int32[F], the sampled Function<Second,Bin> (:555), clamped to [0, num_FRAMES] here exactly as the reference does
// (:556, sic).  The reference ranks with std::sort (:570-571), which is not stable: the order of bins with EQUAL |m| is
// implementation defined.  Checker and device rank equal magnitudes by ascending bin -- one of the orders std::sort may
// produce.  (NaN magnitudes break the comparator's strict weak ordering in the reference: undefined there; here |m| is
// ranked by its bit pattern, so NaN counts as louder than everything.)
int oracle_n_loudest_partials( const float * pv_mf, int ch, int64_t F, int bins, const int32_t * n, int remove, float * out_mf )
	{
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::vector<std::pair<uint32_t, int>> order( bins );
	for( int c = 0; c < ch; ++c )
		for( int64_t frame = 0; frame < F; ++frame )
			{
			const MF * row = in + ( int64_t( c ) * F + frame ) * bins;
			MF * orow = out + ( int64_t( c ) * F + frame ) * bins;
			const int64_t nf = std::clamp<int64_t>( n[frame], 0, F );
			for( int bin = 0; bin < bins; ++bin )
				{
				uint32_t bits; std::memcpy( &bits, &row[bin].m, 4 );
				order[bin] = { bits & 0x7fffffffu, bin };
				}
			std::stable_sort( order.begin(), order.end(), []( const auto & a, const auto & b ){ return a.first > b.first; } );
			for( int rank = 0; rank < bins; ++rank )                                // :575-584
				{
				const int actualBin = order[rank].second;
				const bool keep = remove ? ( rank >= nf ) : ( rank < nf );
				orow[actualBin] = keep ? row[actualBin] : MF{ 0.0f, row[actualBin].f };
				}
			}
	return 0;
	}

// PV::desample, PVModify.cpp:445-511.  ratio: float[F][bins] (:450), clamped per use (:468).  out is cleared (:453):
// frames from the last selected frame on stay zero.
int oracle_desample( const float * pv_mf, int ch, int64_t F, int bins, const float * ratio, int interp_kind, float * out_mf )
	{
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::memset( out, 0, sizeof( MF ) * size_t( ch ) * F * bins );
	for( int c = 0; c < ch; ++c )
		for( int bin = 0; bin < bins; ++bin )
			{
			std::vector<int64_t> selected;
			float accum = 1;                                                        // :461
			for( int64_t frame = 0; frame < F; ++frame )                            // :464-475
				{
				const float factor_c = std::clamp( ratio[frame * bins + bin], 0.0f, 1.0f );
				accum += factor_c;
				if( accum >= 1.0f ) { selected.push_back( frame ); accum -= 1.0f; }
				}
			if( selected.size() < 2 ) continue;                                     // :482
			for( size_t i = 0; i + 1 < selected.size(); ++i )                       // :483-506
				{
				const int64_t lFrame = selected[i], rFrame = selected[i + 1];
				const MF lMF = in[( int64_t( c ) * F + lFrame ) * bins + bin];
				const MF rMF = in[( int64_t( c ) * F + rFrame ) * bins + bin];
				for( int64_t frame = lFrame; frame < rFrame; ++frame )
					{
					const float mix = interpolate( interp_kind, float( frame - lFrame ) / ( rFrame - lFrame ) );   // :491
					const float w0 = ( 1.0f - mix ) * lMF.m;
					const float w1 = mix * rMF.m;
					out[( int64_t( c ) * F + frame ) * bins + bin] = { w0 + w1, w0 > w1 ? lMF.f : rMF.f };       // :494-498
					}
				}
			}
	return 0;
	}

// PV::time_extrapolate, PVModify.cpp:607-666, after its input validation (:612-624, done by the caller):
// start_frame < end_frame, Fo = end_frame + extFrames, interp_samples: float[Fo - start_frame] sampled by the caller as
// the reference does (:631-633).  The reference reads frame `end_frame` without a bounds check (:649; end_frame equals
// num_frames for the default end_time): undefined there, so the caller clamps end_frame to F-1.
int oracle_time_extrapolate( const float * pv_mf, int ch, int64_t F, int bins, float sr, int64_t start_frame, int64_t end_frame, int64_t Fo,
	const float * interp_samples, float * out_mf )
	{
	const int dft = ( bins - 1 ) * 2;
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::memset( out, 0, sizeof( MF ) * size_t( ch ) * Fo * bins );                // :628
	for( int c = 0; c < ch; ++c )
		{
		std::memcpy( out + int64_t( c ) * Fo * bins, in + int64_t( c ) * F * bins, sizeof( MF ) * size_t( start_frame ) * bins );   // :638
		for( int64_t frame = start_frame; frame < Fo; ++frame )                    // :641-663
			{
			const float mix = interp_samples[frame - start_frame];
			MF * orow = out + ( int64_t( c ) * Fo + frame ) * bins;
			for( int bin = 0; bin < bins; ++bin )
				{
				const MF leftMF  = in[( int64_t( c ) * F + start_frame ) * bins + bin];
				const MF rightMF = in[( int64_t( c ) * F + end_frame ) * bins + bin];
				const int32_t rightBinShift = to_int( float( bin ) - frequency_to_bin( rightMF.f, sr, dft ) );       // :654
				const MF extrapMF = { std::abs( ( 1.0f - mix ) * leftMF.m + mix * rightMF.m ),
				                      ( 1.0f - mix ) * leftMF.f + mix * rightMF.f };                                 // :656-657
				const int32_t extrapBin = to_int( frequency_to_bin( extrapMF.f, sr, dft ) + float( rightBinShift ) );   // :658
				if( extrapBin < 0 || extrapBin >= bins ) continue;
				MF & outMF = orow[extrapBin];
				if( extrapMF.m > outMF.m ) outMF = extrapMF;                        // :663-664
				}
			}
		}
	return 0;
	}

} // extern "C"
