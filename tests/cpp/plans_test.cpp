// plans_test.cpp -- the host arithmetic that decides which dft sizes the chirp-z and the residue-pair kernels serve (flan_amd/csrc/bs_plan.h), checked
// over EVERY even size up to 2^20 on the CPU: a plan's factors multiply to what the kernels transform, the strides and magic numbers the passes divide
// by are right for every index they meet, and the two families never claim the same size.
#include "../../flan_amd/csrc/bs_plan.h"
#include <cstdio>
#include <cstdlib>
using namespace flanhip;

static int fails = 0;
#define CHECK( c, ... ) do { if( !( c ) ) { if( fails++ < 20 ) { std::printf( "FAILED %s:%d: ", __FILE__, __LINE__ ); std::printf( __VA_ARGS__ ); std::printf( "\n" ); } } } while( 0 )

template<class Plan> static void check_passes( const Plan & f, int M, int dft, bool pow2 = true )
	{
	long prod = 1;
	for( int i = 0, NS = 1; i < f.npass; NS *= f.radix[i], ++i )
		{
		const int r = f.radix[i];
		CHECK( r == 8 || r == 4 || r == 2 || ( !pow2 && ( r == 3 || r == 5 || r == 7 || r == 11 || r == 13 ) ), "dft %d: radix %d", dft, r );
		CHECK( !pow2 || i > 0 || r == 8, "dft %d: the first pass is a radix-8 one", dft );
		CHECK( f.stride[i] == M / ( NS * r ), "dft %d pass %d: stride", dft, i );
		if( NS > 1 )
			for( int j = 0; j < M / r; j += ( M / r > 4096 ? 7 : 1 ) )                  // j / NS through the magic number, as mr_pass does it
				{
				const unsigned q = unsigned( ( (unsigned long long) (unsigned) j * f.magic[i] ) >> 32 );
				CHECK( int( q ) == j / NS, "dft %d pass %d: %d / %d by magic = %u", dft, i, j, NS, q );
				}
		prod *= r;
		}
	CHECK( prod == M, "dft %d: radices multiply to %ld, not %d", dft, prod, M );
	}

int main()
	{
	int n_bs = 0, n_big = 0, n_mixed = 0, n_used = 0;
	for( int dft = 4; dft <= ( 1 << 20 ); dft += 2 )
		{
		BsPlan bs{}; BigPlan big{};
		const bool is_bs = bs_make_plan( dft, &bs );
		const bool is_big = dft > 16384 && big_make_plan( dft, 4096, &big );
		const int C = dft / 2;
		if( is_bs )
			{
			++n_bs;
			CHECK( bs.C == C && bs.M >= 2 * C - 1 && bs.M <= BSG_MAX_M && ( bs.M & ( bs.M - 1 ) ) == 0 && bs.M < 4 * C, "dft %d: M = %d", dft, bs.M );
			CHECK( ( bs.glob != 0 ) == ( bs.M > BS_MAX_M ), "dft %d: M = %d, glob %d", dft, bs.M, bs.glob );
			CHECK( C >= BS_MIN_C && !bs_has_small_factors_only( C ), "dft %d is no chirp-z size", dft );
			CHECK( bs.sign_c == ( ( C & 1 ) ? -1.0f : 1.0f ), "dft %d: sign", dft );
			if( dft % 97 == 0 || dft < 20000 ) check_passes( bs, bs.M, dft );
			}
		if( is_big )
			{
			++n_big;
			CHECK( big.C == C && big.C1 * big.C2 == C && big.C2 <= BIG_MAX_C2, "dft %d: C1 x C2 = %d x %d", dft, big.C1, big.C2 );
			CHECK( big.C1 >= 2 && big.C1 <= BIG_MAX_C1 && big.P == big.C1 / 2 + 1, "dft %d: C1 = %d, P = %d", dft, big.C1, big.P );
			if( !big.mixed )
				{
				CHECK( big.C2 >= BIG_MIN_C2 && ( big.C2 & ( big.C2 - 1 ) ) == 0, "dft %d: C2 = %d", dft, big.C2 );
				CHECK( ( C / big.C2 ) % 2 == 1 || big.C2 == BIG_MAX_C2, "dft %d: C2 = %d is not the largest power of two", dft, big.C2 );
				}
			else
				{
				// round 6: the largest smooth divisor up to 4096 that leaves C1 <= 256, for the sizes the power-of-two rule does not serve
				++n_mixed;
				int p2 = 1; while( C % ( p2 * 2 ) == 0 && p2 * 2 <= BIG_MAX_C2 ) p2 *= 2;
				CHECK( p2 < BIG_MIN_C2 || C / p2 > BIG_MAX_C1, "dft %d: mixed although 2^k = %d serves it", dft, p2 );
				CHECK( big.C2 >= BIG_MIN_C2_MIXED && bs_has_small_factors_only( big.C2 ), "dft %d: C2 = %d", dft, big.C2 );
				for( int d = big.C2 + 1; d <= BIG_MAX_C2; ++d ) if( C % d == 0 && C / d >= 2 && bs_has_small_factors_only( d ) ) { CHECK( false, "dft %d: C2 = %d although %d divides", dft, big.C2, d ); break; }
				// every in-place pass holds at most MR_POINTS = 16 points per thread of 512 (pv_kernels_mr.h: mr_pass)
				for( int i = 0; i < big.fft.npass; ++i )
					{
					const int r = big.fft.radix[i], per_thread = ( big.C2 / r + 511 ) / 512;
					CHECK( per_thread <= ( 16 + r - 1 ) / r, "dft %d: pass %d of radix %d holds %d butterflies per thread", dft, i, r, per_thread );
					}
				}
			long prod = 1; for( int i = 0; i < big.fft.npass; ++i ) prod *= big.fft.radix[i];
			CHECK( prod == big.C2 && big.fft.M == big.C2, "dft %d: radices multiply to %ld, not %d", dft, prod, big.C2 );
			CHECK( big.N1 == ( 2048 + big.C2 - 1 ) / big.C2 && big.limit == ( 2048 < big.C2 ? 2048 : big.C2 ), "dft %d: a window of 4096 samples = 2048 points in segments of C2 = %d", dft, big.C2 );
			if( dft % 1024 == 0 || ( big.mixed && dft % 100 == 0 ) ) check_passes( big.fft, big.C2, dft, !big.mixed );
			}
		BsPlan used{};
		const bool in_use = bs_plan_in_use( dft, &used );
		CHECK( in_use == ( is_bs && !( bs.glob && is_big ) ), "dft %d: chirp-z plan in use %d, plan %d, glob %d, residue pairs %d", dft, int( in_use ), int( is_bs ), bs.glob, int( is_big ) );
		if( in_use ) ++n_used;
		CHECK( !( is_bs && !bs.glob && is_big ), "dft %d claimed twice", dft );
		}
	// windows: segments and the limit of the first pass
	for( int W : { 2, 5, 4096, 4097, 8192, 8193, 12000, 32768 } )
		{
		BigPlan big{};
		CHECK( big_make_plan( 32768, W, &big ), "dft 32768, window %d", W );
		const int half = ( W + 1 ) / 2;
		CHECK( big.N1 == ( half + 4095 ) / 4096 && big.limit == ( half < 4096 ? half : 4096 ), "window %d: N1 = %d, limit = %d", W, big.N1, big.limit );
		}
	BigPlan big{}; BsPlan bs{};
	CHECK( big_make_plan( 20000, 4096, &big ) && big.mixed && big.C2 == 2500 && big.C1 == 4, "20000 = 2 x 4 x 2500 (its half holds 2^4 only: mixed)" );
	CHECK( big_make_plan( 44100, 4096, &big ) && big.mixed && big.C2 == 3675 && big.C1 == 6, "44100 = 2 x 6 x 3675" );
	CHECK( big_make_plan( 32768, 4096, &big ) && !big.mixed && big.C2 == 4096, "32768 stays a power-of-two plan" );
	CHECK( !big_make_plan( 2 * 10007, 4096, &big ), "2 x 10007: a prime half" );
	CHECK( !bs_make_plan( 3000, &bs ) && bs_make_plan( 2998, &bs ) && bs.M == 4096, "3000 is smooth, 2998 = 2 x 1499 is not" );
	CHECK( bs_make_plan( 8186, &bs ) && bs.M == 8192 && !bs.glob && bs_make_plan( 8198, &bs ) && bs.M == 16384 && bs.glob, "M = 8192 is the last LDS layout; 8198 = 2 x 4099 runs in device memory" );
	CHECK( bs_plan_in_use( 9998, &bs ) && bs.M == 16384 && bs_plan_in_use( 30002, &bs ) && bs.M == 32768, "9998 and 30002 (no residue-pair plan): in device memory" );
	CHECK( !bs_plan_in_use( 180360, &bs ) && bs_make_plan( 180360, &bs ), "180360 = 2 x 167 x 540 has a residue-pair plan: that comes first" );
	CHECK( !bs_make_plan( 2 * 131101, &bs ), "2 x 131101: M would be 2^19" );
	std::printf( "%d chirp-z plans (%d in use), %d sizes above 16384 with residue pairs (%d of them mixed-radix); %s\n", n_bs, n_used, n_big, n_mixed, fails ? "FAILED" : "PASSED" );
	return fails ? 1 : 0;
	}
