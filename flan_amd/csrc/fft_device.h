// fft_device.h -- in-LDS Stockham FFT for gfx950: one 64-lane wavefront per transform, or a team of several (TEAM template parameter).
//
// Replaces the FFTW3f plans of the reference (FFTHelper.cpp:16-48: fftwf_plan_dft_r2c_1d / c2r_1d).
// A real transform of size N is done as a complex transform of C = N/2 points plus a split/merge step.
//
// Layout: the C complex points live in a wave-private LDS buffer, index i stored at PAD(i) = i + (i >> 4)
// (one cf of padding per 16 keeps the radix-16 scatter writes of the first pass on distinct banks).
// Each pass is in place: every lane first reads ALL of its butterflies' inputs into registers, then writes.
// A wavefront issues its LDS operations in program order, so reads-before-writes within one wave needs no
// barrier, only a compiler fence (wave_sync()).
#pragma once
#include <hip/hip_runtime.h>
#include "cf_type.h"

namespace flanhip {

__device__ __forceinline__ int PAD( int i ) { return i + ( i >> 4 ); }
__host__ __device__ constexpr int padded_len( int n ) { return n + ( n >> 4 ) + 2; }

__device__ __forceinline__ void wave_sync()
	{
	__builtin_amdgcn_fence( __ATOMIC_SEQ_CST, "wavefront" );
	__builtin_amdgcn_wave_barrier();
	}

// a transform shared by TEAM threads: one wavefront (a fence is enough, see above) or a whole block of 64 T threads (a barrier)
template<int TEAM> __device__ __forceinline__ void team_sync()
	{
	if constexpr( TEAM == 64 ) wave_sync(); else __syncthreads();
	}

__device__ __forceinline__ cf mk( float a, float b ) { return cf{ a, b }; }
__device__ __forceinline__ cf pk_fma( cf a, cf b, cf c ) { return __builtin_elementwise_fma( a, b, c ); }   // v_pk_fma_f32
__device__ __forceinline__ cf cadd( cf a, cf b ) { return a + b; }                                          // v_pk_add_f32
__device__ __forceinline__ cf csub( cf a, cf b ) { return a - b; }
// ( a.x b.x - a.y b.y , a.x b.y + a.y b.x ): one packed multiply (operands picked with op_sel), one sign flip, one packed fma
__device__ __forceinline__ cf cmul( cf a, cf b )
	{
	cf t = __builtin_shufflevector( a, a, 1, 1 ) * __builtin_shufflevector( b, b, 1, 0 );    // ( a.y b.y , a.y b.x )
	t.x = -t.x;
	return pk_fma( __builtin_shufflevector( a, a, 0, 0 ), b, t );
	}
// multiply by -i
__device__ __forceinline__ cf mul_mi( cf a ) { return cf{ a.y, -a.x }; }

// cos/sin of 2*pi*k/16, k = 0..7  (twiddles inside the register DFTs)
#define FLANHIP_C1 0.92387953251128675613f  /* cos(pi/8) */
#define FLANHIP_S1 0.38268343236508977173f  /* sin(pi/8) */
#define FLANHIP_SQH  0.70710678118654752440f  /* sqrt(1/2) */

// x * exp(-2*pi*i*K/16)
template<int K> __device__ __forceinline__ cf mul_w16( cf a )
	{
	if constexpr( K == 0 ) return a;
	else if constexpr( K == 4 ) return mul_mi( a );
	else if constexpr( K == 2 ) return mk( ( a.x + a.y ) * FLANHIP_SQH, ( a.y - a.x ) * FLANHIP_SQH );
	else if constexpr( K == 6 ) return mk( ( a.y - a.x ) * FLANHIP_SQH, -( a.x + a.y ) * FLANHIP_SQH );
	else if constexpr( K == 1 ) return cmul( a, mk(  FLANHIP_C1, -FLANHIP_S1 ) );
	else if constexpr( K == 3 ) return cmul( a, mk(  FLANHIP_S1, -FLANHIP_C1 ) );
	else if constexpr( K == 5 ) return cmul( a, mk( -FLANHIP_S1, -FLANHIP_C1 ) );
	else                         return cmul( a, mk( -FLANHIP_C1, -FLANHIP_S1 ) ); // K == 7
	}

// Forward R-point DFT (sign -), natural order in and out, fully in registers.
template<int R> struct Reg { cf v[R]; };

template<int R> __device__ __forceinline__ void dft_reg( cf * v );

template<> __device__ __forceinline__ void dft_reg<1>( cf * ) {}
template<> __device__ __forceinline__ void dft_reg<2>( cf * v )
	{
	const cf a = v[0], b = v[1];
	v[0] = cadd( a, b ); v[1] = csub( a, b );
	}
template<> __device__ __forceinline__ void dft_reg<4>( cf * v )
	{
	const cf t0 = cadd( v[0], v[2] ), t1 = csub( v[0], v[2] );
	const cf t2 = cadd( v[1], v[3] ), t3 = mul_mi( csub( v[1], v[3] ) );
	v[0] = cadd( t0, t2 ); v[1] = cadd( t1, t3 ); v[2] = csub( t0, t2 ); v[3] = csub( t1, t3 );
	}
template<> __device__ __forceinline__ void dft_reg<8>( cf * v )
	{
	cf e[4] = { v[0], v[2], v[4], v[6] };
	cf o[4] = { v[1], v[3], v[5], v[7] };
	dft_reg<4>( e ); dft_reg<4>( o );
	const cf o1 = mul_w16<2>( o[1] ), o2 = mul_w16<4>( o[2] ), o3 = mul_w16<6>( o[3] );
	v[0] = cadd( e[0], o[0] ); v[4] = csub( e[0], o[0] );
	v[1] = cadd( e[1], o1 );   v[5] = csub( e[1], o1 );
	v[2] = cadd( e[2], o2 );   v[6] = csub( e[2], o2 );
	v[3] = cadd( e[3], o3 );   v[7] = csub( e[3], o3 );
	}
template<> __device__ __forceinline__ void dft_reg<16>( cf * v )
	{
	// 4 x 4 decomposition: n = 4*n1 + n2, k = k1 + 4*k2
	cf c[4][4];
	#pragma unroll
	for( int n2 = 0; n2 < 4; ++n2 )
		{
		cf t[4] = { v[n2], v[4 + n2], v[8 + n2], v[12 + n2] };
		dft_reg<4>( t );
		#pragma unroll
		for( int k1 = 0; k1 < 4; ++k1 ) c[n2][k1] = t[k1];
		}
	// twiddle W16^(n2*k1)
	c[1][1] = mul_w16<1>( c[1][1] ); c[1][2] = mul_w16<2>( c[1][2] ); c[1][3] = mul_w16<3>( c[1][3] );
	c[2][1] = mul_w16<2>( c[2][1] ); c[2][2] = mul_w16<4>( c[2][2] ); c[2][3] = mul_w16<6>( c[2][3] );
	c[3][1] = mul_w16<3>( c[3][1] ); c[3][2] = mul_w16<6>( c[3][2] );
	// W16^9 = -W16^1
		{
		const cf t = mul_w16<1>( c[3][3] );
		c[3][3] = mk( -t.x, -t.y );
		}
	#pragma unroll
	for( int k1 = 0; k1 < 4; ++k1 )
		{
		cf t[4] = { c[0][k1], c[1][k1], c[2][k1], c[3][k1] };
		dft_reg<4>( t );
		#pragma unroll
		for( int k2 = 0; k2 < 4; ++k2 ) v[k1 + 4 * k2] = t[k2];
		}
	}

// One in-place Stockham pass of radix R over C points, sub-transform length NS on entry.
// tw[i] = exp(-2*pi*i*i/C), i < C (LDS, shared by the block's waves).
template<int C, int R, int NS, int TEAM = 64>
__device__ __forceinline__ void fft_pass( cf * buf, const cf * tw, int lane )
	{
	constexpr int NB = C / R;                       // butterflies in this pass
	constexpr int PER = ( NB + TEAM - 1 ) / TEAM;   // per thread of the team
	cf v[PER][R];
	#pragma unroll
	for( int b = 0; b < PER; ++b )
		{
		const int j = lane + TEAM * b;
		if( NB >= TEAM || j < NB )
			{
			#pragma unroll
			for( int r = 0; r < R; ++r ) v[b][r] = buf[PAD( j + r * NB )];
			}
		}
	team_sync<TEAM>();
	#pragma unroll
	for( int b = 0; b < PER; ++b )
		{
		const int j = lane + TEAM * b;
		if( NB >= TEAM || j < NB )
			{
			const int k = j & ( NS - 1 );
			if constexpr( NS > 1 )
				{
				constexpr int STRIDE = C / ( NS * R );
				#pragma unroll
				for( int r = 1; r < R; ++r ) v[b][r] = cmul( v[b][r], tw[r * k * STRIDE] );
				}
			dft_reg<R>( v[b] );
			const int base = ( j - k ) * R + k;
			#pragma unroll
			for( int r = 0; r < R; ++r ) buf[PAD( base + r * NS )] = v[b][r];
			}
		}
	team_sync<TEAM>();
	}

// Forward complex FFT of C = 2^LOG2C points, in place in `buf` (padded layout), natural order in and out.
template<int LOG2C, int TEAM = 64> __device__ __forceinline__ void fft_forward( cf * buf, const cf * tw, int lane )
	{
	constexpr int C = 1 << LOG2C;
	if constexpr( LOG2C == 4 )       { fft_pass<C, 16, 1, TEAM>( buf, tw, lane ); }
	else if constexpr( LOG2C == 5 )  { fft_pass<C, 8, 1, TEAM>( buf, tw, lane );  fft_pass<C, 4, 8, TEAM>( buf, tw, lane ); }
	else if constexpr( LOG2C == 6 )  { fft_pass<C, 8, 1, TEAM>( buf, tw, lane );  fft_pass<C, 8, 8, TEAM>( buf, tw, lane ); }
	else if constexpr( LOG2C == 7 )  { fft_pass<C, 16, 1, TEAM>( buf, tw, lane ); fft_pass<C, 8, 16, TEAM>( buf, tw, lane ); }
	else if constexpr( LOG2C == 8 )  { fft_pass<C, 16, 1, TEAM>( buf, tw, lane ); fft_pass<C, 16, 16, TEAM>( buf, tw, lane ); }
	else if constexpr( LOG2C == 9 )  { fft_pass<C, 8, 1, TEAM>( buf, tw, lane );  fft_pass<C, 8, 8, TEAM>( buf, tw, lane );   fft_pass<C, 8, 64, TEAM>( buf, tw, lane ); }
	else if constexpr( LOG2C == 10 ) { fft_pass<C, 16, 1, TEAM>( buf, tw, lane ); fft_pass<C, 16, 16, TEAM>( buf, tw, lane ); fft_pass<C, 4, 256, TEAM>( buf, tw, lane ); }
	else if constexpr( LOG2C == 11 ) { fft_pass<C, 16, 1, TEAM>( buf, tw, lane ); fft_pass<C, 16, 16, TEAM>( buf, tw, lane ); fft_pass<C, 8, 256, TEAM>( buf, tw, lane ); }
	else                             { fft_pass<C, 16, 1, TEAM>( buf, tw, lane ); fft_pass<C, 16, 16, TEAM>( buf, tw, lane ); fft_pass<C, 16, 256, TEAM>( buf, tw, lane ); }
	}

} // namespace flanhip
