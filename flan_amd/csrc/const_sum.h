// const_sum.h -- the running fp32 sum of a constant without running it: s_0 = c, s_k = RN( s_{k-1} + c ), k = 1 .. t, for any t in a few dozen steps.
//
// PV::stretch integrates its factor down the frames of every bin in fp32, one addition per frame in frame order (PV/PVModify.cpp:376-378), and a
// CONSTANT factor -- stretch by 2: the everyday call -- makes every column the same sequence.  Sequential by definition, but not by necessity: while s
// stays inside one binade [ 2^e, 2^(e+1) ) every sum s + c is rounded to the SAME grid (multiples of ulp = 2^(e-23)), s is on that grid, so each
// step adds the same whole number of ulps -- c rounded to that grid (ties: to even, which after the first step means always the same way) -- and
// all those additions are exact.  So: jump through a binade in one multiplication, take the step that crosses into the next binade as a real fp32
// addition, repeat: ~2 iterations per binade.  Checked against the sequential sum for every step of millions of ( c, t ) -- powers of two, ties,
// denormals, sums that stagnate or overflow -- by tools/check_const_sum.cpp, and on the device against the oracle's scan (tests/test_gpu_processors.py).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined( __HIPCC__ )
#define FLANHIP_HD __host__ __device__ __forceinline__
#else
#define FLANHIP_HD inline
#endif

namespace flanhip {

FLANHIP_HD uint32_t cs_bits( float x ) { uint32_t u; memcpy( &u, &x, 4 ); return u; }
FLANHIP_HD float cs_float( uint32_t u ) { float x; memcpy( &x, &u, 4 ); return x; }
// a finite non-negative float as S * 2^ue with ue = max( exponent, -126 ) - 23: S < 2^24, and >= 2^23 unless the value is below 2^-126
FLANHIP_HD void cs_split( float x, uint32_t & S, int & ue )
	{
	const uint32_t b = cs_bits( x ) & 0x7FFFFFFFu;
	const uint32_t E = b >> 23, M = b & 0x7FFFFFu;
	S = E ? ( M | 0x800000u ) : M;
	ue = ( E ? int( E ) : 1 ) - 150;
	}
FLANHIP_HD float cs_join( uint32_t S, int ue )                                   // S <= 2^24, the value finite
	{
	if( S >= 0x1000000u ) { S >>= 1; ue += 1; }                                    // ( exactly 2^24 )
	if( S < 0x800000u ) return cs_float( S );                                      // below 2^-126: ue is -149 and S the denormal's own bits
	return cs_float( ( uint32_t( ue + 150 ) << 23 ) | ( S & 0x7FFFFFu ) );
	}

// s_t of the recurrence above
FLANHIP_HD float const_running_sum( float c, uint64_t t )
	{
	const uint32_t cb = cs_bits( c );
	if( ( cb & 0x7F800000u ) == 0x7F800000u || ( cb & 0x7FFFFFFFu ) == 0u ) return c;          // NaN, +-Inf, +-0: every sum is c again
	const float a = cs_float( cb & 0x7FFFFFFFu );                                // |c|: rounding to nearest is symmetric
	uint32_t Mc; int uec;
	cs_split( a, Mc, uec );
	float s = a;
	uint64_t n = t;
	while( n > 0 )
		{
		uint32_t S; int ue;
		cs_split( s, S, ue );
		if( ( cs_bits( s ) & 0x7F800000u ) == 0x7F800000u ) break;                  // overflowed: Inf + c stays Inf
		const int shift = ue - uec;                                                // s >= |c|: never negative
		uint32_t q = 0, r_cmp = 0;                                                 // c = ( q + fraction ) ulp; r_cmp: 0 fraction below a half, 1 a half exactly, 2 above
		if( shift == 0 ) q = Mc;
		else if( shift <= 24 )
			{
			q = Mc >> shift;
			const uint32_t r = Mc & ( ( 1u << shift ) - 1u ), half = 1u << ( shift - 1 );
			r_cmp = r < half ? 0u : r == half ? 1u : 2u;
			}
		const uint32_t TOP = 0x1000000u;
		if( uint64_t( S ) + q >= TOP )
			{
			s = s + a;                                                              // the exact sum leaves the binade: a real addition rounds it on the next grid
			n -= 1;
			continue;
			}
		if( r_cmp == 1u && ( ( S + q ) & 1u ) != ( q & 1u ) )
			{
			S += q + ( ( S + q ) & 1u );                                            // the first tie of a run may go the other way (to an even S); the rest go one way
			n -= 1;
			if( n == 0 || S >= TOP || uint64_t( S ) + q >= TOP ) { s = cs_join( S, ue ); continue; }
			}
		const uint32_t delta = r_cmp == 0u ? q : r_cmp == 2u ? q + 1u : q + ( q & 1u );
		if( delta == 0u ) break;                                                   // c is below half an ulp of s: the sum stands still from here on
		uint32_t m = ( TOP - 1u - q - S ) / delta + 1u;                             // steps whose exact sum stays inside the binade (24-bit operands)
		if( uint64_t( m ) > n ) m = uint32_t( n );
		S += m * delta;                                                            // ( <= 2^24: the last such step may land on the binade's end exactly )
		n -= m;
		s = cs_join( S, ue );
		}
	return ( cb & 0x80000000u ) ? -s : s;
	}

// The same walk recorded once as a short list of RUNS -- ( first step, value after it, ulps added per step ) -- so that a kernel finds any
// step's value with one search and one multiplication: the host walks (microseconds), the device looks up.  A real (binade-crossing) step, the
// point where the sum stands still and an overflow are runs with delta 0.
struct ConstSumRun { uint64_t t_first; uint32_t S; uint32_t delta; int32_t ue; uint32_t value_bits; };     // delta 0: the value is value_bits from t_first on
struct ConstSumRuns { ConstSumRun run[128]; int count; uint32_t sign; float c; };
// runs covering steps 0 .. t_max of |c|'s sum; false if they do not fit (the caller then evaluates const_running_sum per element)
inline bool const_sum_runs( float c, uint64_t t_max, ConstSumRuns * out )
	{
	const uint32_t cb = cs_bits( c );
	out->count = 0; out->sign = cb & 0x80000000u; out->c = c;
	auto push = [&]( uint64_t t_first, uint32_t S, uint32_t delta, int ue, float value ) -> bool
		{
		if( out->count >= 128 ) return false;
		out->run[out->count++] = ConstSumRun{ t_first, S, delta, int32_t( ue ), cs_bits( value ) };
		return true;
		};
	if( ( cb & 0x7F800000u ) == 0x7F800000u || ( cb & 0x7FFFFFFFu ) == 0u ) return push( 0, 0, 0, 0, cs_float( cb & 0x7FFFFFFFu ) );
	const float a = cs_float( cb & 0x7FFFFFFFu );
	uint32_t Mc; int uec;
	cs_split( a, Mc, uec );
	float s = a;
	uint64_t t = 0;                                                                // s is the value at step t
	if( !push( 0, 0, 0, 0, s ) ) return false;
	while( t < t_max )
		{
		uint32_t S; int ue;
		cs_split( s, S, ue );
		if( ( cs_bits( s ) & 0x7F800000u ) == 0x7F800000u ) return true;            // Inf from here on: the last run says so
		const int shift = ue - uec;
		uint32_t q = 0, r_cmp = 0;
		if( shift == 0 ) q = Mc;
		else if( shift <= 24 )
			{
			q = Mc >> shift;
			const uint32_t r = Mc & ( ( 1u << shift ) - 1u ), half = 1u << ( shift - 1 );
			r_cmp = r < half ? 0u : r == half ? 1u : 2u;
			}
		const uint32_t TOP = 0x1000000u;
		if( uint64_t( S ) + q >= TOP )
			{
			s = s + a; t += 1;
			if( !push( t, 0, 0, 0, s ) ) return false;
			continue;
			}
		if( r_cmp == 1u && ( ( S + q ) & 1u ) != ( q & 1u ) )
			{
			S += q + ( ( S + q ) & 1u ); t += 1;
			s = cs_join( S, ue );
			if( !push( t, 0, 0, 0, s ) ) return false;
			if( t >= t_max || S >= TOP || uint64_t( S ) + q >= TOP ) continue;
			}
		const uint32_t delta = r_cmp == 0u ? q : r_cmp == 2u ? q + 1u : q + ( q & 1u );
		if( delta == 0u ) return push( t, 0, 0, 0, s );                            // stands still from here on
		uint64_t m = ( TOP - 1u - q - S ) / delta + 1u;
		if( m > t_max - t ) m = t_max - t;
		if( !push( t + 1, S + delta, delta, ue, 0.0f ) ) return false;              // steps t + 1 .. t + m: S + k delta
		S += uint32_t( m * delta ); t += m;
		s = cs_join( S, ue );
		}
	return true;
	}
// the value at step t (0 <= t <= the t_max the runs were made for): the last run that starts at or before t, by bisection
FLANHIP_HD float const_sum_lookup( const ConstSumRuns & r, uint64_t t )
	{
	int lo = 0, hi = r.count;                                                      // run[lo].t_first <= t < run[hi].t_first (run[0] starts at step 0)
	while( hi - lo > 1 )
		{
		const int mid = ( lo + hi ) >> 1;
		if( r.run[mid].t_first <= t ) lo = mid; else hi = mid;
		}
	const ConstSumRun & u = r.run[lo];
	const float v = u.delta ? cs_join( u.S + uint32_t( t - u.t_first ) * u.delta, u.ue ) : cs_float( u.value_bits );
	return r.sign ? -v : v;
	}

// ... and for a kernel ARGUMENT: the sum at every K-th step, K a power of two chosen so that they fit -- a block reads ONE of them (kernel arguments
// sit behind a long-latency path: a search through the runs from inside a kernel was a chain of seven such reads, ~7 us) and takes the remaining
// steps, fewer than K, as real additions.
struct ConstSumMarks { float c; int log2K; int count; float at[960]; };        // at[k] = s_( k K )
inline bool const_sum_marks( float c, uint64_t t_max, ConstSumMarks * out )
	{
	static thread_local ConstSumRuns runs;
	if( !const_sum_runs( c, t_max, &runs ) ) return false;
	out->c = c; out->log2K = 4;
	while( ( t_max >> out->log2K ) + 1 > 960 ) ++out->log2K;
	out->count = int( ( t_max >> out->log2K ) + 1 );
	for( int k = 0; k < out->count; ++k ) out->at[k] = const_sum_lookup( runs, uint64_t( k ) << out->log2K );
	return true;
	}

} // namespace flanhip
