// bs_plan.h -- which dft sizes run the chirp-z (Bluestein) kernels of pv_kernels_bs.h and the residue-pair kernels of pv_kernels_big.h, and their pass lists (host and device agree through the
// struct; core.hip builds the tables from it, conversions.hip picks the kernels by it).
//
// The reference hands ANY size to FFTW (FFTHelper.cpp:16-26: fftwf_plan_dft_r2c_1d / c2r_1d), which serves sizes with a large prime factor
// in O( N log N ) too.  Here: half the size C = N / 2 (the real transform is a complex one of C points plus the split) NOT a product of
// 2 ... 13 (those run pv_kernels_mr.h), 64 <= C; the C-point transform as a circular convolution of length M = the power of two >= 2 C - 1:
//   Z[k] = conj( w[k] ) sum_n ( z[n] conj( w[n] ) ) w[k - n],   w[n] = exp( + pi i n^2 / C )        ( n k = ( n^2 + k^2 - ( k - n )^2 ) / 2 )
// Two layouts: M <= 4096 with two LDS buffers (ping-pong passes, C <= 2048: dft sizes up to 4096); M = 8192 with one buffer, passes in place and
// their twiddles read through L1 (C <= 4096: dft sizes up to 8192).  Round 6, a third one (`glob`): M = 16384 ... 2^18 (C up to 131072: dft sizes up to 262144) with
// both buffers and what crosses frames in a stretch of device memory per block (ping-pong passes through L2) -- for the sizes above 8192 that nothing else serves,
// which ran the direct sums (O( window x bins ) per frame: ( 9998, 2499, 9998 ) 150 ms for 8 ch x 60 s).  Below C = 64, and above 262144 where neither this nor
// the residue pairs of pv_kernels_big.h apply, the direct sums of pv_kernels_any.h stay.
#pragma once
#include <cstdint>
#include <initializer_list>

namespace flanhip {

constexpr int BS_MIN_C = 64, BS_MAX_M = 8192, BS_PP_MAX_M = 4096, BS_MAX_PASSES = 6;
constexpr int BSG_MAX_M = 1 << 18;           // six radix-8 passes; j / NS by the passes' magic numbers stays exact ( ( M / 8 )^2 < 2^32 ), strides fit 16 bits

struct BsPlan
	{
	int C, M, npass;
	int win_lds;                                 // synthesis: the scaled window in LDS
	int glob;                                    // M above 8192: buffers and state in device memory (BsTables::scratch), ping-pong passes
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
	if( M > BSG_MAX_M ) return false;
	BsPlan pl{};
	pl.C = C; pl.M = M;
	pl.glob = M > BS_MAX_M ? 1 : 0;
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

// ---- sizes above 16384 (pv_kernels_big.h): half the size C = C1 x C2, C1 <= 256 residues of a C2-point transform in LDS ---------------------------------
// C2 = the largest power of two in C up to 4096 where that is at least 1024 and leaves C1 <= 256 (32768, 65536, 24576 ...: radix 8 / 4 / 2 passes, the
// round-5 kernels); otherwise (round 6: 20000 = 2 x 2^4 5^4, 44100, 48000, 100000 ...) the largest divisor of C between 256 and 4096 that is a product of
// 2 ... 13 and leaves 2 <= C1 <= 256 -- the same kernels with pv_kernels_mr.h's odd-radix passes (`mixed`).  Sizes whose half has no such divisor (a prime
// factor above 13 in every one: 2 x 10007 ...) stay on the direct sums.
constexpr int BIG_MAX_C1 = 256, BIG_MIN_C2 = 1024, BIG_MAX_C2 = 4096, BIG_MIN_C2_MIXED = 256, BIG_MAX_PASSES = 12;

struct BigFft
	{
	int M, npass;
	unsigned char radix[BIG_MAX_PASSES];         // C2 = product; 8 / 4 / 2, then 13 / 11 / 7 / 5 / 3
	unsigned magic[BIG_MAX_PASSES];              // (MrPlan::magic, ::stride)
	unsigned short stride[BIG_MAX_PASSES];
	};

struct BigPlan
	{
	int C, C1, C2, P;        // P = C1 / 2 + 1 units per chain
	int N1;                  // segments of C2 complex points the window reaches into ( ceil( ceil( W / 2 ) / C2 ) )
	int limit;               // complex points of a segment that can be non-zero ( min( C2, ceil( W / 2 ) ) )
	int mixed;               // C2 is no power of two: odd-radix passes, bins guarded ( C2 < 512 Q ); 2: a radix 11 or 13 among them (an instantiation of their own)
	BigFft fft;              // the C2-point transform's passes
	};

inline bool big_fft_plan( int C2, BigFft * f )
	{
	*f = BigFft{};
	f->M = C2;
	int rest = C2, n = 0;
	auto take = [&]( int r ) { while( rest % r == 0 && n < BIG_MAX_PASSES ) { f->radix[n++] = (unsigned char) r; rest /= r; } };
	take( 8 ); take( 4 ); take( 2 ); take( 13 ); take( 11 ); take( 7 ); take( 5 ); take( 3 );
	if( rest != 1 ) return false;
	f->npass = n;
	for( int i = 0, NS = 1; i < n; NS *= f->radix[i], ++i )
		{
		f->magic[i] = NS > 1 ? unsigned( ( uint64_t( 1 ) << 32 ) / unsigned( NS ) ) + 1u : 0u;
		f->stride[i] = (unsigned short) ( C2 / ( NS * f->radix[i] ) );
		}
	return true;
	}

inline bool big_make_plan( int dft, int W, BigPlan * out )
	{
	if( dft < 4 || dft % 2 ) return false;
	const int C = dft / 2;
	int C2 = 1;
	while( C % ( C2 * 2 ) == 0 && C2 * 2 <= BIG_MAX_C2 ) C2 *= 2;
	bool mixed = false;
	if( C2 < BIG_MIN_C2 || C / C2 > BIG_MAX_C1 )
		{
		mixed = true;
		C2 = 0;
		for( int d = BIG_MAX_C2; d >= BIG_MIN_C2_MIXED; --d )
			if( C % d == 0 && C / d >= 2 && C / d <= BIG_MAX_C1 && bs_has_small_factors_only( d ) ) { C2 = d; break; }
		if( !C2 ) return false;
		}
	const int C1 = C / C2;
	if( C1 < 2 || C1 > BIG_MAX_C1 ) return false;
	BigPlan pl{};
	pl.C = C; pl.C1 = C1; pl.C2 = C2; pl.P = C1 / 2 + 1;
	pl.mixed = mixed ? 1 : 0;
	if( mixed && ( C2 % 11 == 0 || C2 % 13 == 0 ) ) pl.mixed = 2;
	const int half = ( W + 1 ) / 2;
	pl.N1 = ( half + C2 - 1 ) / C2;
	pl.limit = half < C2 ? half : C2;
	if( !big_fft_plan( C2, &pl.fft ) ) return false;
	*out = pl;
	return true;
	}

// Which sizes the chirp-z kernels serve: every chirp-z size up to 8192; above, the `glob` layout for what the residue pairs do not serve (they come first above 16384)
inline bool bs_plan_in_use( int dft, BsPlan * out )
	{
	BsPlan pl{};
	if( !bs_make_plan( dft, &pl ) ) return false;
	BigPlan big{};
	if( pl.glob && dft > 16384 && big_make_plan( dft, 2, &big ) ) return false;
	*out = pl;
	return true;
	}

} // namespace flanhip
