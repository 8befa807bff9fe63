// bs_plan.h -- which dft sizes run the chirp-z (Bluestein) kernels of pv_kernels_bs.h and the residue-pair kernels of pv_kernels_big.h, and their pass lists (host and device agree through the
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
#include <initializer_list>

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

// ---- sizes above 16384 (pv_kernels_big.h): half the size C = C1 x C2, C2 = the largest power of two in it up to 4096 (at least 1024), C1 <= 256 -------
constexpr int BIG_MAX_C1 = 256, BIG_MIN_C2 = 1024, BIG_MAX_C2 = 4096;

struct BigPlan
	{
	int C, C1, C2, P;        // P = C1 / 2 + 1 units per chain
	int N1;                  // segments of C2 complex points the window reaches into ( ceil( ceil( W / 2 ) / C2 ) )
	int limit;               // complex points of a segment that can be non-zero ( min( C2, ceil( W / 2 ) ) )
	BsPlan fft;              // the C2-point transform's passes ( M = C2 )
	};

inline bool big_make_plan( int dft, int W, BigPlan * out )
	{
	if( dft < 4 || dft % 2 ) return false;
	const int C = dft / 2;
	int C2 = 1;
	while( C % ( C2 * 2 ) == 0 && C2 * 2 <= BIG_MAX_C2 ) C2 *= 2;
	if( C2 < BIG_MIN_C2 ) return false;
	const int C1 = C / C2;
	if( C1 < 2 || C1 > BIG_MAX_C1 ) return false;
	BigPlan pl{};
	pl.C = C; pl.C1 = C1; pl.C2 = C2; pl.P = C1 / 2 + 1;
	const int half = ( W + 1 ) / 2;
	pl.N1 = ( half + C2 - 1 ) / C2;
	pl.limit = half < C2 ? half : C2;
	BsPlan & f = pl.fft;
	f.C = C2; f.M = C2; f.sign_c = 1.0f; f.win_lds = 0;
	int rest = C2, n = 0;
	auto take = [&]( int r ) { while( rest % r == 0 && n < BS_MAX_PASSES ) { f.radix[n++] = (unsigned char) r; rest /= r; } };
	take( 8 ); take( 4 ); take( 2 );
	if( rest != 1 ) return false;
	f.npass = n;
	for( int i = 0, NS = 1; i < n; NS *= f.radix[i], ++i )
		{
		f.magic[i] = NS > 1 ? unsigned( ( uint64_t( 1 ) << 32 ) / unsigned( NS ) ) + 1u : 0u;
		f.stride[i] = (unsigned short) ( C2 / ( NS * f.radix[i] ) );
		}
	*out = pl;
	return true;
	}

} // namespace flanhip
