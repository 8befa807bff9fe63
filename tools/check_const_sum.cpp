// tools/check_const_sum.cpp -- flan_amd/csrc/const_sum.h against the sum it replaces, step by step:  g++ -O2 -ffp-contract=off -Iflan_amd/csrc tools/check_const_sum.cpp -o /tmp/ccs && /tmp/ccs
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include "const_sum.h"
using namespace flanhip;

static long check( float c, uint64_t steps, long & bad )
	{
	volatile float s = c;
	long n = 0;
	static ConstSumRuns runs;
	const bool have_runs = const_sum_runs( c, steps, &runs );
	if( !have_runs ) { std::printf( "runs of c=%a do not fit\n", c ); ++bad; }
	for( uint64_t t = 0; t <= steps; ++t )
		{
		if( t ) s = s + c;
		// every step for short runs, a thinning sample beyond
		if( t < 3000 || ( t % 997 ) == 0 || t == steps )
			{
			const float g = const_running_sum( c, t );
			const float sv = s;
			++n;
			if( have_runs )
				{
				const float h = const_sum_lookup( runs, t );
				if( cs_bits( h ) != cs_bits( sv ) && !( h != h && sv != sv ) )
					{
					if( bad < 20 ) std::printf( "MISMATCH (runs) c=%a t=%llu: sequential %a, looked up %a\n", c, (unsigned long long) t, sv, h );
					++bad;
					}
				}
			if( cs_bits( g ) != cs_bits( sv ) && !( g != g && sv != sv ) )
				{
				if( bad < 20 ) std::printf( "MISMATCH c=%a t=%llu: sequential %a, closed form %a\n", c, (unsigned long long) t, sv, g );
				++bad;
				}
			}
		}
	return n;
	}

int main( int argc, char ** argv )
	{
	const int rounds = argc > 1 ? std::atoi( argv[1] ) : 3000;
	std::mt19937 rng( 12345 );
	long bad = 0, n = 0;
	std::vector<float> special = { 2.0f, 1.0f, 0.5f, 3.0f, 1.5f, 0.75f, 1.3f, 0.1f, 0.7f, 2.5f, 1e-3f, 1e3f, 0x1p-126f, 0x1p-127f, 0x1p-149f, 0x1.8p-149f, 0x3p-149f, 0x1.fffffep-1f,
	                               0x1.000002p0f, 0x1.fffffep127f, 0x1p127f, 1e38f, 3e38f, 0x1.000002p-126f, 0x1.8p-120f, -2.0f, -1.3f, -0x1p-140f, 0.0f, -0.0f, INFINITY, -INFINITY, NAN,
	                               16777216.0f, 8388608.0f, 8388607.5f, 0x1.555556p-2f, 0x1.99999ap-4f };
	for( float c : special ) n += check( c, 200000, bad );
	for( int i = 0; i < rounds; ++i )
		{
		uint32_t u = rng();
		if( i % 4 == 0 ) u = ( u & 0x807FFFFFu ) | ( ( 100u + rng() % 56u ) << 23 );      // everyday magnitudes
		if( i % 7 == 0 ) u &= 0xFFFF0000u;                                                 // short mantissas: ties
		if( i % 11 == 0 ) u &= 0xFF800000u;                                                // powers of two
		float c; memcpy( &c, &u, 4 );
		n += check( c, 30000 + rng() % 100000, bad );
		}
	// a few long runs (hours of audio at small hops: tens of millions of frames)
	for( float c : { 2.0f, 1.3f, 0.333f, 1.0f, 0.5f, 7.77f } ) n += check( c, 40000000ull, bad );
	std::printf( "%ld checks, %ld mismatches\n", n, bad );
	return bad ? 1 : 0;
	}
