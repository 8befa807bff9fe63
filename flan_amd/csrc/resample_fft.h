// resample_fft.h -- the 2:1 block convolver of Audio::resample as fp64 overlap-save FFT convolution: k_resample_ols3 (below; what runs) and its
// predecessor k_resample_ols2 (kept behind the resample_direct = 2 hook for A/B).
//
// Reference: r8brain/CDSPBlockConvolver.h:242-344 convolves by FFT overlap-save (for 96 -> 48 kHz: 1621 taps, blocks of 2476 new input
// samples, fft 4096, inverse of half the size because only every second output is wanted).  The direct 1621-tap FIR of k_resample_down<2>
// spends 1621 fp64 FMAs per output; this kernel ~190 fp64 operations:
//   * TWO consecutive blocks per workgroup, packed as real and imaginary part of one complex sequence  z = s_b + i s_(b+1)  (the filter is
//     real, so its action on both is one complex multiplication per bin; no real-transform split / merge passes);
//   * Z = FFT_4096( z ) as three radix-16 Stockham passes: 256 threads x 16 points, the first pass straight from global memory, the third
//     one ending in registers with thread j holding Z[j + 256 r];
//   * W = Z H ( H = FFT( taps ) / 4096, from the host in long double );  the decimated sequence c[2u] has the spectrum
//     E[k] = W[k] + W[k + 2048]: both terms are in the same thread (r and r + 8) -- and E[j + 256 r], r < 8, is exactly what the first
//     (radix-8) pass of a 2048-point transform wants;
//   * inverse transform = conj( FFT_2048( conj E ) ) as 8 x 16 x 16; thread j < 128 ends with c[2 ( j + 128 r )] of block b in the real and of
//     block b + 1 in the (negated) imaginary part; the first fl2 outputs of each are the overlap-save wrap-around and are dropped.
// Outputs equal the direct sum to ~1e-15 relative before the rounding to float (the direct kernel stays as the checker-order path for short
// inputs and for the fp64 streams between the stages of a chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flanhip {

struct cd { double x, y; };

__device__ __forceinline__ cd cd_add( cd a, cd b ) { return cd{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ cd cd_sub( cd a, cd b ) { return cd{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ cd cd_mul( cd a, cd b ) { return cd{ __builtin_fma( a.x, b.x, -( a.y * b.y ) ), __builtin_fma( a.x, b.y, a.y * b.x ) }; }
__device__ __forceinline__ cd cd_mul_mi( cd a ) { return cd{ a.y, -a.x }; }                                    // * ( -i )

#define FLANHIP_D_C1  0.92387953251128675613   /* cos( pi / 8 ) */
#define FLANHIP_D_S1  0.38268343236508977173   /* sin( pi / 8 ) */
#define FLANHIP_D_SQH 0.70710678118654752440   /* sqrt( 1 / 2 ) */

// x * exp( -2 pi i K / 16 )
template<int K> __device__ __forceinline__ cd cd_mul_w16( cd a )
	{
	if constexpr( K == 0 ) return a;
	else if constexpr( K == 4 ) return cd_mul_mi( a );
	else if constexpr( K == 2 ) return cd{ ( a.x + a.y ) * FLANHIP_D_SQH, ( a.y - a.x ) * FLANHIP_D_SQH };
	else if constexpr( K == 6 ) return cd{ ( a.y - a.x ) * FLANHIP_D_SQH, -( a.x + a.y ) * FLANHIP_D_SQH };
	else if constexpr( K == 1 ) return cd_mul( a, cd{  FLANHIP_D_C1, -FLANHIP_D_S1 } );
	else if constexpr( K == 3 ) return cd_mul( a, cd{  FLANHIP_D_S1, -FLANHIP_D_C1 } );
	else if constexpr( K == 5 ) return cd_mul( a, cd{ -FLANHIP_D_S1, -FLANHIP_D_C1 } );
	else                         return cd_mul( a, cd{ -FLANHIP_D_C1, -FLANHIP_D_S1 } );
	}

__device__ __forceinline__ void cd_dft4( cd & a, cd & b, cd & c, cd & d )
	{
	const cd t0 = cd_add( a, c ), t1 = cd_sub( a, c ), t2 = cd_add( b, d ), t3 = cd_mul_mi( cd_sub( b, d ) );
	a = cd_add( t0, t2 ); b = cd_add( t1, t3 ); c = cd_sub( t0, t2 ); d = cd_sub( t1, t3 );
	}

// forward 8-point transform, natural order in and out
__device__ __forceinline__ void cd_dft8( cd * v )
	{
	cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
	cd_dft4( e0, e1, e2, e3 ); cd_dft4( o0, o1, o2, o3 );
	o1 = cd_mul_w16<2>( o1 ); o2 = cd_mul_w16<4>( o2 ); o3 = cd_mul_w16<6>( o3 );
	v[0] = cd_add( e0, o0 ); v[4] = cd_sub( e0, o0 );
	v[1] = cd_add( e1, o1 ); v[5] = cd_sub( e1, o1 );
	v[2] = cd_add( e2, o2 ); v[6] = cd_sub( e2, o2 );
	v[3] = cd_add( e3, o3 ); v[7] = cd_sub( e3, o3 );
	}

// forward 16-point transform as 4 x 4 ( n = 4 n1 + n2, k = k1 + 4 k2 ), natural order in and out
__device__ __forceinline__ void cd_dft16( cd * v )
	{
	cd c[4][4];
	#pragma unroll
	for( int n2 = 0; n2 < 4; ++n2 )
		{
		cd a = v[n2], b = v[4 + n2], g = v[8 + n2], d = v[12 + n2];
		cd_dft4( a, b, g, d );
		c[n2][0] = a; c[n2][1] = b; c[n2][2] = g; c[n2][3] = d;
		}
	c[1][1] = cd_mul_w16<1>( c[1][1] ); c[1][2] = cd_mul_w16<2>( c[1][2] ); c[1][3] = cd_mul_w16<3>( c[1][3] );
	c[2][1] = cd_mul_w16<2>( c[2][1] ); c[2][2] = cd_mul_w16<4>( c[2][2] ); c[2][3] = cd_mul_w16<6>( c[2][3] );
	c[3][1] = cd_mul_w16<3>( c[3][1] ); c[3][2] = cd_mul_w16<6>( c[3][2] );
		{
		const cd t = cd_mul_w16<1>( c[3][3] );                                   // W16^9 = -W16^1
		c[3][3] = cd{ -t.x, -t.y };
		}
	#pragma unroll
	for( int k1 = 0; k1 < 4; ++k1 )
		{
		cd a = c[0][k1], b = c[1][k1], g = c[2][k1], d = c[3][k1];
		cd_dft4( a, b, g, d );
		v[k1] = a; v[k1 + 4] = b; v[k1 + 8] = g; v[k1 + 12] = d;
		}
	}

// twiddle tables of the four passes that have them, one after the other (cd units)
struct OlsTables
	{
	static constexpr int F1 = 0;                     // [15][16]    exp( -2 pi i r k / 256 )    forward, pass 1
	static constexpr int F2 = F1 + 15 * 16;          // [15][256]   exp( -2 pi i r k / 4096 )   forward, pass 2
	static constexpr int I1 = F2 + 15 * 256;         // [15][8]     exp( -2 pi i r k / 128 )    inverse, pass 1
	static constexpr int I2 = I1 + 15 * 8;           // [15][128]   exp( -2 pi i r k / 2048 )   inverse, pass 2
	static constexpr int LEN = I2 + 15 * 128;
	};

constexpr int OLS_N = 4096, OLS_THREADS = 256;
constexpr int OLS_BUF = OLS_N + OLS_N / 16;            // one pad slot per 16: the radix-16 scatter of a pass lands on distinct banks
__device__ __forceinline__ int ols_pad( int i ) { return i + ( i >> 4 ); }

// out[k] = OutT( sum_m h[2 fl2 - m] x[2k - fl2 + m] ), k < total_out (the sum k_resample_down<2> takes term by term); workgroup w serves the
// output blocks 2w and 2w + 1 of Lo = 2048 - fl2 outputs each
template<typename InT, typename OutT>
__global__ __launch_bounds__( OLS_THREADS ) void k_resample_ols2( const InT * __restrict__ in, int64_t total_in, const cd * __restrict__ spec,
	const cd * __restrict__ tw, int fl2, OutT * __restrict__ out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cd * buf = reinterpret_cast<cd*>( smem );
	const int j = threadIdx.x;
	const int Lo = OLS_N / 2 - fl2, L = 2 * Lo;
	const int64_t k0 = int64_t( 2 * blockIdx.x ) * Lo;                          // first output of the first block
	const int64_t n0 = 2 * k0 - fl2;                                            // its segment starts here; the second block's L samples later
	cd v[16];
	// ---- forward pass 0: radix 16, points j + 256 r straight from the signal (zero outside it, like the direct sum's bounds)
	#pragma unroll
	for( int r = 0; r < 16; ++r )
		{
		const int64_t a = n0 + j + 256 * r, b = a + L;
		v[r].x = ( a >= 0 && a < total_in ) ? double( in[a] ) : 0.0;
		v[r].y = ( b >= 0 && b < total_in ) ? double( in[b] ) : 0.0;
		}
	cd t1[15];
	#pragma unroll
	for( int r = 0; r < 15; ++r ) t1[r] = tw[OlsTables::F1 + r * 16 + ( j & 15 )];
	cd_dft16( v );
	#pragma unroll
	for( int r = 0; r < 16; ++r ) buf[ols_pad( 16 * j + r )] = v[r];
	__syncthreads();
	// ---- forward pass 1: radix 16, sub-transform length 16
	#pragma unroll
	for( int r = 0; r < 16; ++r ) v[r] = buf[ols_pad( j + 256 * r )];
	__syncthreads();
	#pragma unroll
	for( int r = 1; r < 16; ++r ) v[r] = cd_mul( v[r], t1[r - 1] );
	#pragma unroll
	for( int r = 0; r < 15; ++r ) t1[r] = tw[OlsTables::F2 + r * 256 + j];         // pass 2's twiddles travel during the butterflies
	cd_dft16( v );
		{
		const int k = j & 15, base = ( j - k ) * 16 + k;
		#pragma unroll
		for( int r = 0; r < 16; ++r ) buf[ols_pad( base + 16 * r )] = v[r];
		}
	__syncthreads();
	// ---- forward pass 2: radix 16, sub-transform length 256: leaves Z[j + 256 r] in v[r]
	#pragma unroll
	for( int r = 0; r < 16; ++r ) v[r] = buf[ols_pad( j + 256 * r )];
	__syncthreads();
	#pragma unroll
	for( int r = 1; r < 16; ++r ) v[r] = cd_mul( v[r], t1[r - 1] );
	cd_dft16( v );
	// ---- filter, fold to the spectrum of every second output, conjugate for the inverse transform
	cd e[8];
	#pragma unroll
	for( int r = 0; r < 8; ++r )
		{
		const cd a = cd_mul( v[r], spec[j + 256 * r] ), b = cd_mul( v[r + 8], spec[j + 256 * ( r + 8 )] );
		e[r] = cd{ a.x + b.x, -( a.y + b.y ) };
		}
	// ---- inverse pass 0: radix 8 over E[j + 256 r]
	cd_dft8( e );
	#pragma unroll
	for( int r = 0; r < 8; ++r ) buf[ols_pad( 8 * j + r )] = e[r];
	__syncthreads();
	if( j < 128 )
		{
		#pragma unroll
		for( int r = 0; r < 15; ++r ) t1[r] = tw[OlsTables::I1 + r * 8 + ( j & 7 )];
		#pragma unroll
		for( int r = 0; r < 16; ++r ) v[r] = buf[ols_pad( j + 128 * r )];
		}
	__syncthreads();
	if( j < 128 )
		{
		// ---- inverse pass 1: radix 16, sub-transform length 8
		#pragma unroll
		for( int r = 1; r < 16; ++r ) v[r] = cd_mul( v[r], t1[r - 1] );
		#pragma unroll
		for( int r = 0; r < 15; ++r ) t1[r] = tw[OlsTables::I2 + r * 128 + j];
		cd_dft16( v );
		const int k = j & 7, base = ( j - k ) * 16 + k;
		#pragma unroll
		for( int r = 0; r < 16; ++r ) buf[ols_pad( base + 8 * r )] = v[r];
		}
	__syncthreads();
	if( j < 128 )
		{
		// ---- inverse pass 2: radix 16, sub-transform length 128: c[2u] of both blocks, u = j + 128 r
		#pragma unroll
		for( int r = 0; r < 16; ++r ) v[r] = buf[ols_pad( j + 128 * r )];
		#pragma unroll
		for( int r = 1; r < 16; ++r ) v[r] = cd_mul( v[r], t1[r - 1] );
		cd_dft16( v );
		#pragma unroll
		for( int r = 0; r < 16; ++r )
			{
			const int u = j + 128 * r - fl2;                                      // the first fl2 are the wrap-around of the circular convolution
			if( u >= 0 )
				{
				const int64_t ka = k0 + u, kb = ka + Lo;
				if( ka < total_out ) out[ka] = OutT( v[r].x );
				if( kb < total_out ) out[kb] = OutT( -v[r].y );
				}
			}
		}
	}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_resample_ols3: the same convolver with twice the wavefronts per CU.  k_resample_ols2's counters (profiles/r04_resample_sq_counters.txt): a
// wavefront waits 64 % of its life -- on table reads through L2 and on the barriers between six short passes -- with two wavefronts per SIMD
// to cover that (256 threads x 16 points: 214 registers, 68 KB of LDS) and half the workgroup idle through the inverse passes.  Here: 512
// threads x 8 points, radix-8 Stockham passes ( 4096 = 8 x 8 x 8 x 8 forward, 2048 = 4 x 8 x 8 x 8 inverse ), <= 128 registers: four
// wavefronts per SIMD at the same LDS footprint (two workgroups per CU), ~4 200 fp64 wave-instructions per workgroup instead of 5 460, 66
// memory reads per thread instead of 93.  The packing (two blocks as real and imaginary part), the fold E[k] = W[k] + W[k + 2048] and the dropped
// wrap-around are k_resample_ols2's.
struct Ols3Tables
	{
	static constexpr int F1 = 0;                     // [7][8]     exp( -2 pi i r k / 64 )     forward, sub-transforms of 8
	static constexpr int F2 = F1 + 7 * 8;            // [7][64]    exp( -2 pi i r k / 512 )
	static constexpr int F3 = F2 + 7 * 64;           // [7][512]   exp( -2 pi i r k / 4096 )
	static constexpr int I1 = F3 + 7 * 512;          // [7][4]     exp( -2 pi i r k / 32 )     inverse ( 2048 points ), sub-transforms of 4
	static constexpr int I2 = I1 + 7 * 4;            // [7][32]    exp( -2 pi i r k / 256 )
	static constexpr int I3 = I2 + 7 * 32;           // [7][256]   exp( -2 pi i r k / 2048 )
	static constexpr int LEN = I3 + 7 * 256;
	};
constexpr int OLS3_THREADS = 512;
constexpr int OLS3_BUF = OLS_N + OLS_N / 8;            // one pad slot per 8: the radix-8 scatter of a pass lands on distinct banks
__device__ __forceinline__ int ols3_pad( int i ) { return i + ( i >> 3 ); }

// The transforms are IN PLACE -- forward decimation in frequency (natural order in, digit-reversed spectrum out), inverse decimation in time
// (digit-reversed in, natural out) -- so a thread writes the slots it read and ONE barrier separates two passes (a Stockham pass needs two);
// the filter spectrum is stored in the order the forward transform leaves the bins in (spec3[t * 512 + q] = H[rev( q ) + 512 t], rev = the
// three radix-8 digits of q reversed), and the fold pairs ( k, k + 2048 ) are still one thread's: the last forward pass leaves thread q with
// Z[rev( q ) + 512 t], t < 8, and position 4 q + r of the inverse's array is E[rev( q ) + 512 r] -- exactly the order a 4 x 8 x 8 x 8
// decimation-in-time transform of 2048 points starts from.
template<typename InT, typename OutT>
__global__ __launch_bounds__( OLS3_THREADS, 2 ) void k_resample_ols3( const InT * __restrict__ in, int64_t total_in, const cd * __restrict__ spec3,
	const cd * __restrict__ tw, int fl2, OutT * __restrict__ out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	cd * buf = reinterpret_cast<cd*>( smem );
	const int j = threadIdx.x;
	const unsigned ju = threadIdx.x;                                             // (table offsets: 32-bit, no sign extension)
	const int Lo = OLS_N / 2 - fl2, L = 2 * Lo;
	const int64_t k0 = int64_t( 2 * blockIdx.x ) * Lo;                          // first output of the first block
	const int64_t n0 = 2 * k0 - fl2;                                            // its segment starts here; the second block's L samples later
	cd v[8], t[7], h7;
	// ---- forward pass A (blocks of 4096): points j + 512 r straight from the signal (zero outside it, like the direct sum's bounds)
	if( n0 >= 0 && n0 + L + OLS_N <= total_in )                                 // (every workgroup but the first and the last few: no bounds tests)
		{
		const InT * pa = in + n0 + j;
		#pragma unroll
		for( int r = 0; r < 8; ++r ) { v[r].x = double( pa[512 * r] ); v[r].y = double( pa[L + 512 * r] ); }
		}
	else
		{
		#pragma unroll
		for( int r = 0; r < 8; ++r )
			{
			const int64_t a = n0 + j + 512 * r, b = a + L;
			v[r].x = ( a >= 0 && a < total_in ) ? double( in[a] ) : 0.0;
			v[r].y = ( b >= 0 && b < total_in ) ? double( in[b] ) : 0.0;
			}
		}
	#pragma unroll
	for( int r = 0; r < 7; ++r ) t[r] = tw[Ols3Tables::F3 + r * 512 + ju];
	cd_dft8( v );
	#pragma unroll
	for( int r = 1; r < 8; ++r ) v[r] = cd_mul( v[r], t[r - 1] );
	#pragma unroll
	for( int r = 0; r < 7; ++r ) t[r] = tw[Ols3Tables::F2 + r * 64 + ( ju & 63 )];  // (the next pass's twiddles travel across the barrier)
	#pragma unroll
	for( int r = 0; r < 8; ++r ) buf[ols3_pad( j + 512 * r )] = v[r];
	__syncthreads();
	// ---- forward passes B (blocks of 512) and C (blocks of 64): butterfly, then the twiddle, back into the same slots
	#pragma unroll
	for( int pass = 0; pass < 2; ++pass )
		{
		const int base = pass == 0 ? ( j >> 6 ) * 512 + ( j & 63 ) : ( j >> 3 ) * 64 + ( j & 7 ), step = pass == 0 ? 64 : 8;
		#pragma unroll
		for( int r = 0; r < 8; ++r ) v[r] = buf[ols3_pad( base + step * r )];
		cd_dft8( v );
		#pragma unroll
		for( int r = 1; r < 8; ++r ) v[r] = cd_mul( v[r], t[r - 1] );
		if( pass == 0 )
			{
			#pragma unroll
			for( int r = 0; r < 7; ++r ) t[r] = tw[Ols3Tables::F1 + r * 8 + ( ju & 7 )];
			}
		else
			{
			// (pass D has no twiddles: the filter's spectrum travels in their registers)
			#pragma unroll
			for( int r = 0; r < 7; ++r ) t[r] = spec3[ju + 512 * r];
			h7 = spec3[ju + 512 * 7];
			}
		#pragma unroll
		for( int r = 0; r < 8; ++r ) buf[ols3_pad( base + step * r )] = v[r];
		__syncthreads();
		}
	// ---- forward pass D (blocks of 8, no twiddles): leaves Z[rev( j ) + 512 r] in v[r]
	#pragma unroll
	for( int r = 0; r < 8; ++r ) v[r] = buf[ols3_pad( 8 * j + r )];
	cd_dft8( v );
	// ---- filter, fold to the spectrum of every second output, conjugate for the inverse transform; inverse pass 1: radix 4 over E[rev( j ) + 512 r]
	cd e[4];
	#pragma unroll
	for( int r = 0; r < 4; ++r )
		{
		const cd a = cd_mul( v[r], t[r] ), b = cd_mul( v[r + 4], r == 3 ? h7 : t[r + 4] );
		e[r] = cd{ a.x + b.x, -( a.y + b.y ) };
		}
	cd_dft4( e[0], e[1], e[2], e[3] );
	__syncthreads();                                                            // (pass D's reads of other threads' slots are done)
	#pragma unroll
	for( int r = 0; r < 4; ++r ) buf[ols3_pad( 4 * j + r )] = e[r];
	const bool mine = j < 256;                                                  // 256 butterflies of 8 from here on
	if( mine )
		{
		#pragma unroll
		for( int r = 0; r < 7; ++r ) t[r] = tw[Ols3Tables::I1 + r * 4 + ( ju & 3 )];
		}
	__syncthreads();
	// ---- inverse passes 2 (sub-transforms of 4 -> 32) and 3 (32 -> 256): the twiddle, then the butterfly, back into the same slots
	#pragma unroll
	for( int pass = 0; pass < 2; ++pass )
		{
		if( mine )
			{
			const int base = pass == 0 ? ( j >> 2 ) * 32 + ( j & 3 ) : ( j >> 5 ) * 256 + ( j & 31 ), step = pass == 0 ? 4 : 32;
			#pragma unroll
			for( int r = 0; r < 8; ++r ) v[r] = buf[ols3_pad( base + step * r )];
			#pragma unroll
			for( int r = 1; r < 8; ++r ) v[r] = cd_mul( v[r], t[r - 1] );
			#pragma unroll
			for( int r = 0; r < 7; ++r ) t[r] = pass == 0 ? tw[Ols3Tables::I2 + r * 32 + ( ju & 31 )] : tw[Ols3Tables::I3 + r * 256 + ju];
			cd_dft8( v );
			#pragma unroll
			for( int r = 0; r < 8; ++r ) buf[ols3_pad( base + step * r )] = v[r];
			}
		__syncthreads();
		}
	if( mine )
		{
		// ---- inverse pass 4 (256 -> 2048): c[2u] of both blocks, u = j + 256 r
		#pragma unroll
		for( int r = 0; r < 8; ++r ) v[r] = buf[ols3_pad( j + 256 * r )];
		#pragma unroll
		for( int r = 1; r < 8; ++r ) v[r] = cd_mul( v[r], t[r - 1] );
		cd_dft8( v );
		if( k0 + L <= total_out )
			{
			OutT * po = out + k0 + ( j - fl2 );
			#pragma unroll
			for( int r = 0; r < 8; ++r )
				if( j + 256 * r >= fl2 ) { po[256 * r] = OutT( v[r].x ); po[Lo + 256 * r] = OutT( -v[r].y ); }   // the first fl2 are the wrap-around of the circular convolution
			}
		else
			{
			#pragma unroll
			for( int r = 0; r < 8; ++r )
				{
				const int u = j + 256 * r - fl2;
				if( u >= 0 )
					{
					const int64_t ka = k0 + u, kb = ka + Lo;
					if( ka < total_out ) out[ka] = OutT( v[r].x );
					if( kb < total_out ) out[kb] = OutT( -v[r].y );
					}
				}
			}
		}
	}

} // namespace flanhip
