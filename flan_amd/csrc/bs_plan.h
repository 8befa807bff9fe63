// bs_plan.h -- which dft sizes run the chirp-z (Bluestein) kernels of pv_kernels_bs.h, and their pass lists (host and device agree through the
// struct; core.hip builds the tables from it, conversions.hip picks the kernels by it).
//
// The reference hands ANY size to FFTW (FFTHelper.cpp:16-26: fftwf_plan_dft_r2c_1d / c2r_1d), which serves sizes with a large prime factor
// in O( N log N ) too.  Here: half the size C = N / 2 (the real transform is a complex one of C points plus the split) NOT a product of
// 2 ... 13 (those run pv_kernels_mr.h), 64 <= C; the C-point transform as a circular convolution of length M = the power of two >= 2 C - 1:
//   Z[k] = conj( w[k] ) sum_n ( z[n] conj( w[n] ) ) w[k - n],   w[n] = exp( + pi i n^2 / C )        ( n k = ( n^2 + k^2 - ( k - n )^2 ) / 2 )
// Two layouts: M <= 4096 with two LDS buffers (ping-pong passes, C <= 2048: dft sizes up to 4096); M = 8192 with one buffer, passes in place and
// their twiddles read through L1 (C <= 4096: dft sizes up to 8192).  Beyond, and below C = 64, the direct sums of pv_kernels_any.h stay.
#pragma once
#include <cstdint>

namespace flanhip {

constexpr int BS_MIN_C = 64, BS_MAX_M = 8192, BS_PP_MAX_M = 4096, BS_MAX_PASSES = 6;

struct BsPlan
	{
	int C, M, npass;
	int win_lds;                                 // synthesis: the scaled window in LDS
	float sign_c;                                // ( -1 )^C:  w[C - k] = sign_c w[k]
	unsigned char radix[BS_MAX_PASSES];          // M = product; 8s first (the first pass is always a radix-8 one: M >= 128), then 4 / 2 (in place: 16s first)
	unsigned magic[BS_MAX_PASSES];               // floor( 2^32 / NS ) + 1 of the pass (MrPlan::magic)
	unsigned short stride[BS_MAX_PASSES];        // M / ( NS R ) of the pass
	};

inline bool bs_has_small_factors_only( int C )
	{
	for( int r : { 2, 3, 5, 7, 11, 13 } ) while( C % r == 0 ) C /= r;
	return C == 1;
	}

inline bool bs_make_plan( int dft, BsPlan * out )
	{
	if( dft < 4 || dft % 2 ) return false;
	const int C = dft / 2;
	if( C < BS_MIN_C || bs_has_small_factors_only( C ) ) return false;
	int M = 128;
	while( M < 2 * C - 1 ) M *= 2;
	if( M > BS_MAX_M ) return false;
	BsPlan pl{};
	pl.C = C; pl.M = M;
	pl.sign_c = ( C & 1 ) ? -1.0f : 1.0f;
	int rest = M, n = 0;
	auto take = [&]( int r ) { while( rest % r == 0 && n < BS_MAX_PASSES ) { pl.radix[n++] = (unsigned char) r; rest /= r; } };
	take( 8 ); take( 4 ); take( 2 );
	if( rest != 1 ) return false;
	pl.npass = n;
	for( int i = 0, NS = 1; i < n; NS *= pl.radix[i], ++i )
		{
		pl.magic[i] = NS > 1 ? unsigned( ( uint64_t( 1 ) << 32 ) / unsigned( NS ) ) + 1u : 0u;
		pl.stride[i] = (unsigned short) ( M / ( NS * pl.radix[i] ) );
		}
	*out = pl;
	return true;
	}

} // namespace flanhip
