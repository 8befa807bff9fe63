// pv_math.h -- per-bin arithmetic of the phase vocoder with the reference's rounding sequence
// (phase_vocoder.cpp:37-61), written so that the expensive steps cost a few instructions on gfx950.
//
// Everything here is compiled with -ffp-contract=off: a*b+c is two roundings unless fmaf() is written out.
// The helpers are branch-free on purpose: a data-dependent `if` inside per-lane code costs an exec-mask
// save/restore (5-8 scalar instructions) per use and there are ~10 uses per bin; rare cases (operands outside the
// range a helper is exact for) are instead flagged by the caller once per frame and that frame is redone through
// the slow, fully general routines.
#pragma once
#include <hip/hip_runtime.h>
#include "cf_type.h"

namespace flanhip {

// defines.h:44-45: pi = acos(-1.0f) (float), pi2 = pi * 2.0f  -> 6.2831854820251465 as a float, NOT 2*pi.
#define FLANHIP_PI2_F 6.2831854820251465f
#define FLANHIP_PI2_D 6.2831854820251465
#define FLANHIP_RPI2_F 0x1.45f306p-3f       /* RN( 1 / pi2 ) */
#define FLANHIP_PI_F   0x1.921fb6p+1f
#define FLANHIP_PIO2_F 0x1.921fb6p+0f

// x / pi2 with IEEE round-to-nearest semantics in 3 instructions.  q0 = RN(x*rc); r = x - q0*c exactly (fma);
// q = RN(q0 + r*rc) is the correctly rounded quotient for EVERY float with |x| >= 1e-30 -- checked exhaustively
// over all 3.8e9 such floats by tools/check_div_pi2.cpp.  Below 1e-30 the residual is subnormal and the last place
// of the quotient may differ from the hardware division; every use in this path either rounds the quotient to an
// integer (0 either way) or adds it to a bin frequency it is >= 20 orders of magnitude below.
__device__ __forceinline__ float div_pi2( float x )
	{
	const float q0 = x * FLANHIP_RPI2_F;
	const float r = __builtin_fmaf( -q0, FLANHIP_PI2_F, x );
	return __builtin_fmaf( r, FLANHIP_RPI2_F, q0 );
	}

// Division by a run-time constant c (the analysis rate).  The same 3-instruction form is correctly rounded for most, not
// all, divisors; the host proves it for the c at hand by trying every float |x| >= 1e-30 on the device once per process
// (core.hip: get_div_plan) and kernels take the hardware division when the proof failed.
struct DivC { float c, rc; int exact; };
__device__ __forceinline__ float div_c( float x, DivC d )
	{
	if( d.exact )
		{
		const float q0 = x * d.rc;
		const float r = __builtin_fmaf( -q0, d.c, x );
		return __builtin_fmaf( r, d.rc, q0 );
		}
	return x / d.c;
	}

// atan2f for finite operands: one reciprocal + Newton step for min/max, an 8-coefficient odd minimax polynomial on
// [0,1] (tools/fit_atan.py: max error 1.8 ulp, mean 0.37 ulp), octant fix-ups with the float constants pi/2 and pi,
// sign of y.  atan2( +-0, +-0 ) follows C99 ( +-0 for x = +0, +-pi for x = -0 ).  Operands are expected finite and, when not
// zero, inside [2^-126, 2^126] (any audio spectrum); two infinite operands give NaN where libm gives a multiple of pi/4.
__device__ __forceinline__ float atan2_fast( float y, float x )
	{
	const float ax = __builtin_fabsf( x ), ay = __builtin_fabsf( y );
	// the clamp makes 0/0 come out as q = 0 (so atan2(+-0, +-0) needs no special case) and keeps rcp finite
	const float mx = __builtin_fmaxf( __builtin_fmaxf( ax, ay ), 0x1p-126f ), mn = __builtin_fminf( ax, ay );
	const float r = __builtin_amdgcn_rcpf( mx );
	const float q0 = mn * r;
	const float q = __builtin_fmaf( __builtin_fmaf( -q0, mx, mn ), r, q0 );
	const float u = q * q;
	float p = 0x1.7ec8b6p-9f;
	p = __builtin_fmaf( p, u, -0x1.0c272ap-6f );
	p = __builtin_fmaf( p, u, 0x1.61f9a0p-5f );
	p = __builtin_fmaf( p, u, -0x1.3554c4p-4f );
	p = __builtin_fmaf( p, u, 0x1.b4e022p-4f );
	p = __builtin_fmaf( p, u, -0x1.230ab4p-3f );
	p = __builtin_fmaf( p, u, 0x1.9978eep-3f );
	p = __builtin_fmaf( p, u, -0x1.5554dcp-2f );
	float a = __builtin_fmaf( q * u, p, q );
	a = ( ay > ax ) ? FLANHIP_PIO2_F - a : a;
	const bool xneg = __float_as_int( x ) < 0;                 // sign bit, so that -0 counts
	a = xneg ? FLANHIP_PI_F - a : a;
	return __builtin_copysignf( a, y );
	}

// ---- several bins at a time ------------------------------------------------------------------------------------------------
// The per-bin arithmetic is long DEPENDENT chains (Horner polynomials, Newton steps).  A gfx950 SIMD can start an fp32
// VALU instruction every 2 cycles, but a dependent one only every ~5-7 (measured, tools/ubench/valu_issue.hip), and with 2
// wavefronts per SIMD nothing else fills the gap: hipcc schedules the chain of ONE bin (or pair) back to back with s_nops.
// So the helpers below are written on native float vectors of N = 4 or 8 bins: every step is N/2 independent packed
// instructions (v_pk_fma_f32 ...), which is exactly the instruction-level parallelism the pipeline needs.  Same
// operations and rounding as the scalar helpers above; rcp, sqrt, round, compares and selects stay one per bin.
typedef float v4f __attribute__(( ext_vector_type( 4 ) ));
typedef float v8f __attribute__(( ext_vector_type( 8 ) ));
template<class V> struct vec_traits { static constexpr int N = int( sizeof( V ) / sizeof( float ) ); };
template<class V> __device__ __forceinline__ V vfma( V a, V b, V c ) { return __builtin_elementwise_fma( a, b, c ); }
template<class V> __device__ __forceinline__ V vsplat( float v ) { V r; for( int i = 0; i < vec_traits<V>::N; ++i ) r[i] = v; return r; }
template<class V> __device__ __forceinline__ V vabs( V a ) { return __builtin_elementwise_abs( a ); }
template<class V> __device__ __forceinline__ V vmax( V a, V b ) { return __builtin_elementwise_max( a, b ); }
template<class V> __device__ __forceinline__ V vmin( V a, V b ) { return __builtin_elementwise_min( a, b ); }
// The same N-bin streams as N separate floats (pv_kernels_v3.h): a native vector that lives across the frame loop (the previous phases ARE
// the last frame's phase vector) is a register TUPLE to the allocator -- four or eight consecutive aligned registers -- and under the
// register caps of the dft 1024 / 512 kernels tuples are what it fails to place (it spills whole tuples to park one element).  The helpers
// below take either kind: same operations, same order.
template<int N_> struct FA
	{
	float v[N_];
	__device__ __forceinline__ float & operator[]( int i ) { return v[i]; }
	__device__ __forceinline__ float operator[]( int i ) const { return v[i]; }
	};
template<int N> __device__ __forceinline__ FA<N> operator+( FA<N> a, FA<N> b ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = a.v[i] + b.v[i]; return r; }
template<int N> __device__ __forceinline__ FA<N> operator-( FA<N> a, FA<N> b ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = a.v[i] - b.v[i]; return r; }
template<int N> __device__ __forceinline__ FA<N> operator*( FA<N> a, FA<N> b ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = a.v[i] * b.v[i]; return r; }
template<int N> __device__ __forceinline__ FA<N> operator-( FA<N> a ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = -a.v[i]; return r; }
template<int N> __device__ __forceinline__ FA<N> vfma( FA<N> a, FA<N> b, FA<N> c ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = __builtin_fmaf( a.v[i], b.v[i], c.v[i] ); return r; }
template<int N> __device__ __forceinline__ FA<N> vabs( FA<N> a ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = __builtin_fabsf( a.v[i] ); return r; }
template<int N> __device__ __forceinline__ FA<N> vmax( FA<N> a, FA<N> b ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = __builtin_fmaxf( a.v[i], b.v[i] ); return r; }
template<int N> __device__ __forceinline__ FA<N> vmin( FA<N> a, FA<N> b ) { FA<N> r; for( int i = 0; i < N; ++i ) r.v[i] = __builtin_fminf( a.v[i], b.v[i] ); return r; }

template<class V> __device__ __forceinline__ V div_pi2_v( V x )
	{
	const V rc = vsplat<V>( FLANHIP_RPI2_F ), c = vsplat<V>( FLANHIP_PI2_F );
	const V q0 = x * rc;
	const V r = vfma( -q0, c, x );
	return vfma( r, rc, q0 );
	}
template<class V> __device__ __forceinline__ V div_c_v( V x, DivC d )
	{
	if( d.exact )
		{
		const V rc = vsplat<V>( d.rc ), c = vsplat<V>( d.c );
		const V q0 = x * rc;
		const V r = vfma( -q0, c, x );
		return vfma( r, rc, q0 );
		}
	V q;
	#pragma unroll
	for( int i = 0; i < vec_traits<V>::N; ++i ) q[i] = x[i] / d.c;
	return q;
	}
// the same, element by element with the constants as scalar operands: a vsplat of a run-time constant makes the compiler hold one SGPR
// copy per element (8 x 2 for a DivC in an 8-bin stream), and the dft 2048 kernels have no SGPRs to spare
template<class V> __device__ __forceinline__ V div_c_each( V x, DivC d )
	{
	V q;
	if( d.exact )
		{
		#pragma unroll
		for( int i = 0; i < vec_traits<V>::N; ++i )
			{
			const float q0 = x[i] * d.rc;
			const float r = __builtin_fmaf( -q0, d.c, x[i] );
			q[i] = __builtin_fmaf( r, d.rc, q0 );
			}
		return q;
		}
	#pragma unroll
	for( int i = 0; i < vec_traits<V>::N; ++i ) q[i] = x[i] / d.c;
	return q;
	}
template<class V> __device__ __forceinline__ V round_v( V x )
	{
	V r;
	#pragma unroll
	for( int i = 0; i < vec_traits<V>::N; ++i ) r[i] = roundf( x[i] );
	return r;
	}
template<class V> __device__ __forceinline__ V atan2_fast_v( V y, V x )
	{
	constexpr int N = vec_traits<V>::N;
	const V ax = vabs( x ), ay = vabs( y );
	const V mx = vmax( vmax( ax, ay ), vsplat<V>( 0x1p-126f ) );
	const V mn = vmin( ax, ay );
	V r;
	#pragma unroll
	for( int i = 0; i < N; ++i ) r[i] = __builtin_amdgcn_rcpf( mx[i] );
	const V q0 = mn * r;
	const V q = vfma( vfma( -q0, mx, mn ), r, q0 );
	const V u = q * q;
	V p = vsplat<V>( 0x1.7ec8b6p-9f );
	p = vfma( p, u, vsplat<V>( -0x1.0c272ap-6f ) );
	p = vfma( p, u, vsplat<V>( 0x1.61f9a0p-5f ) );
	p = vfma( p, u, vsplat<V>( -0x1.3554c4p-4f ) );
	p = vfma( p, u, vsplat<V>( 0x1.b4e022p-4f ) );
	p = vfma( p, u, vsplat<V>( -0x1.230ab4p-3f ) );
	p = vfma( p, u, vsplat<V>( 0x1.9978eep-3f ) );
	p = vfma( p, u, vsplat<V>( -0x1.5554dcp-2f ) );
	V a = vfma( q * u, p, q );
	const V a1 = vsplat<V>( FLANHIP_PIO2_F ) - a;
	#pragma unroll
	for( int i = 0; i < N; ++i ) a[i] = ay[i] > ax[i] ? a1[i] : a[i];
	const V a2 = vsplat<V>( FLANHIP_PI_F ) - a;
	#pragma unroll
	for( int i = 0; i < N; ++i ) a[i] = __builtin_copysignf( __float_as_int( x[i] ) < 0 ? a2[i] : a[i], y[i] );
	return a;
	}
template<class V> __device__ __forceinline__ V magnitude_scaled_v( V re, V im )
	{
	constexpr int N = vec_traits<V>::N;
	const V a = vmax( vabs( re ), vabs( im ) );
	int e[N];
	V rs, is;
	#pragma unroll
	for( int i = 0; i < N; ++i )
		{
		e[i] = __builtin_amdgcn_frexp_expf( a[i] );
		rs[i] = __builtin_ldexpf( re[i], -e[i] );
		is[i] = __builtin_ldexpf( im[i], -e[i] );
		}
	const V s2 = vfma( rs, rs, is * is );
	V m;
	#pragma unroll
	for( int i = 0; i < N; ++i ) m[i] = __builtin_ldexpf( __builtin_amdgcn_sqrtf( s2[i] ), e[i] );
	return m;
	}
template<class V> __device__ __forceinline__ void sincos_fast_v( V x, V & s, V & c )
	{
	constexpr int N = vec_traits<V>::N;
	// k = rint( x 2 / pi ) by the add-and-subtract of 1.5 2^23 (round to nearest even, exactly rint for |x 2 / pi| < 2^22): two full-rate
	// additions where v_rndne_f32 and v_cvt_i32_f32 are half-rate instructions on gfx950, and the sum's low mantissa bits ARE k's low bits
	const V kf = x * vsplat<V>( 0x1.45f306p-1f );
	const V km = kf + vsplat<V>( 0x1.8p+23f );
	const V k = km - vsplat<V>( 0x1.8p+23f );
	V r = vfma( -k, vsplat<V>( 0x1.921fb6p+0f ), x );
	r = vfma( -k, vsplat<V>( -0x1.777a5cp-25f ), r );
	r = vfma( -k, vsplat<V>( -0x1.ee59dap-50f ), r );
	const V r2 = r * r;
	V sp = vsplat<V>( 0x1.6cd1e4p-19f );
	sp = vfma( sp, r2, vsplat<V>( -0x1.a00f80p-13f ) );
	sp = vfma( sp, r2, vsplat<V>( 0x1.111108p-7f ) );
	sp = vfma( sp, r2, vsplat<V>( -0x1.555556p-3f ) );
	const V sr = vfma( r * r2, sp, r );
	V cp = vsplat<V>( 0x1.99eb7cp-16f );
	cp = vfma( cp, r2, vsplat<V>( -0x1.6c0c34p-10f ) );
	cp = vfma( cp, r2, vsplat<V>( 0x1.55554ap-5f ) );
	cp = vfma( cp, r2, vsplat<V>( -0x1.000000p-1f ) );
	const V cr = vfma( cp, r2, vsplat<V>( 1.0f ) );
	#pragma unroll
	for( int i = 0; i < N; ++i )
		{
		const int q = __float_as_int( km[i] );                                          // k modulo 2^22 in the low bits
		const float ss = ( q & 1 ) ? cr[i] : sr[i], cc = ( q & 1 ) ? sr[i] : cr[i];
		s[i] = ( q & 2 ) ? -ss : ss;
		c[i] = ( ( q + 1 ) & 2 ) ? -cc : cc;
		}
	}

// phase_vocoder.cpp:57-59:  phase_buffer += term; if( phase_buffer > pi2 ) phase_buffer = fmod( phase_buffer, pi2 ).
// fmod is exact; with P = pi2 * 2^j (exactly representable: pi2 is a 24-bit constant) and ph < 2^29 P the quotient
// q = floor(ph/P) is below 2^29, q*P is exact in double and so is ph - q*P (fma), so floor + fma + one correction
// reproduces fmod( ph, P ) without a division.  fold_phase_fast is the single stage j = 0 (ph < 3e9, i.e. every
// phase a real PV produces); fold_phase_any chains stages for arbitrarily large ph.  NaN stays NaN, +inf becomes NaN
// (as fmod does).
__device__ __forceinline__ double fold_stage( double ph, double P, double rP )
	{
	const double q = __builtin_floor( ph * rP );
	double r = __builtin_fma( -q, P, ph );
	r = ( r < 0.0 ) ? r + P : r;
	r = ( r >= P ) ? r - P : r;
	return r;
	}
__device__ __forceinline__ double fold_phase_fast( double ph )
	{
	const double r = fold_stage( ph, FLANHIP_PI2_D, 1.0 / FLANHIP_PI2_D );
	return ( ph > FLANHIP_PI2_D ) ? r : ph;
	}
#define FLANHIP_FOLD_FAST_LIMIT 3.0e9
// The fold of the synthesis frame loop: floor, clamp, fma -- four fp64 instructions instead of fold_phase_fast's fifteen.  q = floor( ph / P )
// clamped at 0 leaves ph <= P (negative ones included) as it is, like phase_vocoder.cpp:59; above P, ph - q P is exact.  The two
// corrections of fold_stage only ever fire when ph * rP rounds across a whole number (ph within ~1e-16 q of a multiple of P): there this
// returns the representative just below 0 instead of the one just below P -- the same angle to 1e-15 rad, in about one step in 1e13.
__device__ __forceinline__ double fold_phase_loop( double ph )
	{
	const double q = __builtin_fmax( __builtin_floor( ph * ( 1.0 / FLANHIP_PI2_D ) ), 0.0 );
	return __builtin_fma( -q, FLANHIP_PI2_D, ph );
	}
__device__ __noinline__ double fold_phase_any( double ph )
	{
	if( !( ph > FLANHIP_PI2_D ) ) return ph;
	if( ph > 1.7e308 ) return __builtin_nan( "" );
	int e; (void) frexp( ph, &e );
	for( int j = ( ( e > 3 ? e - 3 : 0 ) / 28 ) * 28; j >= 0; j -= 28 )
		{
		const double P = ldexp( FLANHIP_PI2_D, j );
		if( ph >= P ) ph = fold_stage( ph, P, 1.0 / P );
		}
	return ph;
	}

// magnitude with the operands pre-scaled by a power of two (exact), so that the squares neither overflow nor underflow
// for any finite input: |z| = 2^e * sqrt( (re 2^-e)^2 + (im 2^-e)^2 ), e = exponent of max(|re|,|im|).
__device__ __forceinline__ float magnitude_scaled( float re, float im )
	{
	const float a = __builtin_fmaxf( __builtin_fabsf( re ), __builtin_fabsf( im ) );
	const int e = __builtin_amdgcn_frexp_expf( a );
	const float rs = __builtin_ldexpf( re, -e ), is = __builtin_ldexpf( im, -e );
	return __builtin_ldexpf( __builtin_amdgcn_sqrtf( __builtin_fmaf( rs, rs, is * is ) ), e );
	}

// sin and cos of a float angle beyond the range of sincos_fast: the same polynomials after a Cody-Waite reduction
// carried out in double (valid while the quotient is an exact double integer with room to spare: |x| < 2^45; a float
// that large has an ulp of 4e6 radians).
__device__ __noinline__ float2 sincos_wide( float x )
	{
	const double xd = double( x );
	const double k = __builtin_rint( xd * 0.63661977236758134308 );
	double r = __builtin_fma( -k, 1.57079632679489655800e+00, xd );
	r = __builtin_fma( -k, 6.12323399573676603587e-17, r );
	r = __builtin_fma( -k, -1.49738490485916983693e-33, r );
	const float rf = float( r );
	const int q = int( k - 4.0 * __builtin_floor( k * 0.25 ) );
	const float r2 = rf * rf;
	float sp = 0x1.6cd1e4p-19f;
	sp = __builtin_fmaf( sp, r2, -0x1.a00f80p-13f );
	sp = __builtin_fmaf( sp, r2, 0x1.111108p-7f );
	sp = __builtin_fmaf( sp, r2, -0x1.555556p-3f );
	const float sr = __builtin_fmaf( rf * r2, sp, rf );
	float cp = 0x1.99eb7cp-16f;
	cp = __builtin_fmaf( cp, r2, -0x1.6c0c34p-10f );
	cp = __builtin_fmaf( cp, r2, 0x1.55554ap-5f );
	cp = __builtin_fmaf( cp, r2, -0x1.000000p-1f );
	const float cr = __builtin_fmaf( cp, r2, 1.0f );
	const float ss = ( q & 1 ) ? cr : sr;
	const float cc = ( q & 1 ) ? sr : cr;
	return make_float2( ( q & 2 ) ? -ss : ss, ( ( q + 1 ) & 2 ) ? -cc : cc );
	}
#define FLANHIP_SINCOS_FAST_LIMIT 0x1p22f

// sin and cos of a float angle for |x| < 2^22 (the folded phase lives in [0, 2 pi]): Cody-Waite reduction by pi/2 in
// three parts, minimax polynomials on [-pi/4, pi/4] (tools/fit_sincos.py: max abs error 7e-8 up to 8192, and below the
// argument's own ulp beyond).  Branch-free; larger arguments are the caller's business (sincos_wide).
__device__ __forceinline__ void sincos_fast( float x, float & s, float & c )
	{
	const float km = x * 0x1.45f306p-1f + 0x1.8p+23f;                                 // rint( x 2 / pi ) + 1.5 2^23 (see sincos_fast_v)
	const float k = km - 0x1.8p+23f;
	float r = __builtin_fmaf( -k, 0x1.921fb6p+0f, x );
	r = __builtin_fmaf( -k, -0x1.777a5cp-25f, r );
	r = __builtin_fmaf( -k, -0x1.ee59dap-50f, r );
	const float r2 = r * r;
	float sp = 0x1.6cd1e4p-19f;
	sp = __builtin_fmaf( sp, r2, -0x1.a00f80p-13f );
	sp = __builtin_fmaf( sp, r2, 0x1.111108p-7f );
	sp = __builtin_fmaf( sp, r2, -0x1.555556p-3f );
	const float sr = __builtin_fmaf( r * r2, sp, r );
	float cp = 0x1.99eb7cp-16f;
	cp = __builtin_fmaf( cp, r2, -0x1.6c0c34p-10f );
	cp = __builtin_fmaf( cp, r2, 0x1.55554ap-5f );
	cp = __builtin_fmaf( cp, r2, -0x1.000000p-1f );
	const float cr = __builtin_fmaf( cp, r2, 1.0f );
	const int q = __float_as_int( km );
	const float ss = ( q & 1 ) ? cr : sr;
	const float cc = ( q & 1 ) ? sr : cr;
	s = ( q & 2 ) ? -ss : ss;
	c = ( ( q + 1 ) & 2 ) ? -cc : cc;
	}

// roundf for every float (ties away from zero): trunc( x + copysign( prev( 0.5 ), x ) )
template<class V> __device__ __forceinline__ V round_half_away_v( V x )
	{
	constexpr int N = vec_traits<V>::N;
	V r;
	#pragma unroll
	for( int i = 0; i < N; ++i ) r[i] = __builtin_truncf( x[i] + __builtin_copysignf( 0x1.fffffep-2f, x[i] ) );
	return r;
	}
__device__ __forceinline__ float round_half_away( float x ) { return __builtin_truncf( x + __builtin_copysignf( 0x1.fffffep-2f, x ) ); }

// the arithmetic behind polar_v once |re|, |im|, their maximum and minimum are known.  CLAMPED: the divisor clamped to 2^-126 and
// p2 = min( ( max 2^126 )^2, 1 ) in place of the 1 under the root (digital silence, see polar_v)
template<bool CLAMPED, class V> __device__ __forceinline__ void polar_tail( V re, V im, V ax, V ay, V mxu, V mn, V & phase, V & mag )
	{
	constexpr int N = vec_traits<V>::N;
	V mx = mxu;
	if constexpr( CLAMPED ) mx = vmax( mxu, vsplat<V>( 0x1p-126f ) );
	V r;
	#pragma unroll
	for( int i = 0; i < N; ++i ) r[i] = __builtin_amdgcn_rcpf( mx[i] );
	const V q0 = mn * r;
	const V q = vfma( vfma( -q0, mx, mn ), r, q0 );
	const V u = q * q;
	V h;
	if constexpr( CLAMPED )
		{
		const V t = mxu * vsplat<V>( 0x1p126f );
		h = vfma( q, q, vmin( t * t, vsplat<V>( 1.0f ) ) );
		}
	else h = vfma( q, q, vsplat<V>( 1.0f ) );
	V p = vsplat<V>( 0x1.7ec8b6p-9f );
	p = vfma( p, u, vsplat<V>( -0x1.0c272ap-6f ) );
	p = vfma( p, u, vsplat<V>( 0x1.61f9a0p-5f ) );
	p = vfma( p, u, vsplat<V>( -0x1.3554c4p-4f ) );
	p = vfma( p, u, vsplat<V>( 0x1.b4e022p-4f ) );
	p = vfma( p, u, vsplat<V>( -0x1.230ab4p-3f ) );
	p = vfma( p, u, vsplat<V>( 0x1.9978eep-3f ) );
	p = vfma( p, u, vsplat<V>( -0x1.5554dcp-2f ) );
	V a = vfma( q * u, p, q );
	const V a1 = vsplat<V>( FLANHIP_PIO2_F ) - a;
	#pragma unroll
	for( int i = 0; i < N; ++i ) a[i] = ay[i] > ax[i] ? a1[i] : a[i];
	const V a2 = vsplat<V>( FLANHIP_PI_F ) - a;
	#pragma unroll
	for( int i = 0; i < N; ++i )
		{
		phase[i] = __builtin_copysignf( __float_as_int( re[i] ) < 0 ? a2[i] : a[i], im[i] );
		mag[i] = mx[i] * __builtin_amdgcn_sqrtf( h[i] );
		}
	}

// phase = atan2( im, re ) exactly as atan2_fast_v; mag = max( |re|, |im| ) * sqrt( 1 + q^2 ), q = min / max (correctly rounded).
// The maximum is clamped to 2^-126 for the reciprocal; a spectrum component below that (a denormal, or zero) keeps its magnitude
// through p2 = ( max 2^126 )^2 < 1 in place of the 1:  mag = 2^-126 sqrt( p2 + q^2 )  (for every other input p2 clamps to exactly 1).
// Is any larger component below 2^-126 (zero or a denormal: digital silence)?  Decided per wavefront, and a BRANCH: for every other input
// the clamp changes nothing and p2 is exactly 1, so the common path carries neither.  (Written as `if( silence ) mx = max( ... )` the
// compiler turns the branch into a v_max and a v_cndmask per bin on the common path -- both half-rate instructions on gfx950,
// profiles/r03_a_issue_model.txt; the asm statement in the rare arm is what keeps it a branch.)
template<class V> __device__ __forceinline__ void polar_v( V re, V im, V & phase, V & mag )
	{
	constexpr int N = vec_traits<V>::N;
	const V ax = vabs( re ), ay = vabs( im );
	const V mxu = vmax( ax, ay );
	const V mn = vmin( ax, ay );
	float tiny = mxu[0];
	#pragma unroll
	for( int i = 1; i < N; ++i ) tiny = __builtin_fminf( tiny, mxu[i] );
	if( __builtin_expect( __any( tiny < 0x1p-126f ), 0 ) )
		{
		asm volatile( "; digital silence" );
		polar_tail<true>( re, im, ax, ay, mxu, mn, phase, mag );
		asm volatile( "; end of the silence arm" );                               // (no tail shared with the common arm: merged tails cost it a register copy per bin)
		}
	else polar_tail<false>( re, im, ax, ay, mxu, mn, phase, mag );
	}


} // namespace flanhip
