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

// |z| = sqrt( re^2 + im^2 ) by one fma and the hardware square root (v_sqrt_f32, 1 ulp): within 1.5 ulp of hypotf for
// max(|re|,|im|) in [1e-18, 1e18]; the caller tracks the largest / smallest operand of a frame and redoes the frame
// with hypotf when it falls outside.
__device__ __forceinline__ float magnitude_fast( float re, float im )
	{
	return __builtin_amdgcn_sqrtf( __builtin_fmaf( re, re, im * im ) );
	}
__device__ __forceinline__ bool magnitude_fast_ok( float amax, float amin_nonzero )
	{
	return amax < 1.0e18f && amin_nonzero > 1.0e-18f;
	}

// atan2f for finite operands: one reciprocal + Newton step for min/max, an 8-coefficient odd minimax polynomial on
// [0,1] (tools/fit_atan.py: max error 1.8 ulp, mean 0.37 ulp), octant fix-ups with the float constants pi/2 and pi,
// sign of y.  atan2( +-0, +-0 ) follows C99 ( +-0 for x = +0, +-pi for x = -0 ).  Infinite operands are not handled
// (a frame that holds one is redone through atan2f by the caller: magnitude_fast_ok() is false for it).
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

// ---- two bins at a time --------------------------------------------------------------------------------------------------
// A plain fp32 VALU instruction occupies its SIMD for 4 cycles whatever it computes; the packed forms (v_pk_fma_f32,
// v_pk_mul_f32, v_pk_add_f32) produce two results in the same slot.  The per-bin arithmetic of two adjacent-register bins is
// therefore evaluated as one `cf` (native 2-float vector) stream: every add / multiply / fma below is a packed
// instruction; only rcp, sqrt, round, compares and selects stay one per bin.  Same operations, same rounding, as the scalar
// helpers above.
__device__ __forceinline__ cf pk_fma2( cf a, cf b, cf c ) { return __builtin_elementwise_fma( a, b, c ); }
__device__ __forceinline__ cf bcast2( float v ) { return cf{ v, v }; }

__device__ __forceinline__ cf div_pi2_2( cf x )
	{
	const cf q0 = x * bcast2( FLANHIP_RPI2_F );
	const cf r = pk_fma2( -q0, bcast2( FLANHIP_PI2_F ), x );
	return pk_fma2( r, bcast2( FLANHIP_RPI2_F ), q0 );
	}
__device__ __forceinline__ cf div_c_2( cf x, DivC d )
	{
	if( d.exact )
		{
		const cf q0 = x * bcast2( d.rc );
		const cf r = pk_fma2( -q0, bcast2( d.c ), x );
		return pk_fma2( r, bcast2( d.rc ), q0 );
		}
	return cf{ x.x / d.c, x.y / d.c };
	}
__device__ __forceinline__ cf atan2_fast_2( cf y, cf x )
	{
	const cf ax = __builtin_elementwise_abs( x ), ay = __builtin_elementwise_abs( y );
	const cf mx = __builtin_elementwise_max( __builtin_elementwise_max( ax, ay ), bcast2( 0x1p-126f ) );
	const cf mn = __builtin_elementwise_min( ax, ay );
	const cf r = cf{ __builtin_amdgcn_rcpf( mx.x ), __builtin_amdgcn_rcpf( mx.y ) };
	const cf q0 = mn * r;
	const cf q = pk_fma2( pk_fma2( -q0, mx, mn ), r, q0 );
	const cf u = q * q;
	cf p = bcast2( 0x1.7ec8b6p-9f );
	p = pk_fma2( p, u, bcast2( -0x1.0c272ap-6f ) );
	p = pk_fma2( p, u, bcast2( 0x1.61f9a0p-5f ) );
	p = pk_fma2( p, u, bcast2( -0x1.3554c4p-4f ) );
	p = pk_fma2( p, u, bcast2( 0x1.b4e022p-4f ) );
	p = pk_fma2( p, u, bcast2( -0x1.230ab4p-3f ) );
	p = pk_fma2( p, u, bcast2( 0x1.9978eep-3f ) );
	p = pk_fma2( p, u, bcast2( -0x1.5554dcp-2f ) );
	cf a = pk_fma2( q * u, p, q );
	const cf a1 = bcast2( FLANHIP_PIO2_F ) - a;
	a = cf{ ay.x > ax.x ? a1.x : a.x, ay.y > ax.y ? a1.y : a.y };
	const cf a2 = bcast2( FLANHIP_PI_F ) - a;
	a = cf{ __float_as_int( x.x ) < 0 ? a2.x : a.x, __float_as_int( x.y ) < 0 ? a2.y : a.y };
	return cf{ __builtin_copysignf( a.x, y.x ), __builtin_copysignf( a.y, y.y ) };
	}
__device__ __forceinline__ cf magnitude_scaled_2( cf re, cf im )
	{
	const cf a = __builtin_elementwise_max( __builtin_elementwise_abs( re ), __builtin_elementwise_abs( im ) );
	const int e0 = __builtin_amdgcn_frexp_expf( a.x ), e1 = __builtin_amdgcn_frexp_expf( a.y );
	const cf rs = cf{ __builtin_ldexpf( re.x, -e0 ), __builtin_ldexpf( re.y, -e1 ) };
	const cf is = cf{ __builtin_ldexpf( im.x, -e0 ), __builtin_ldexpf( im.y, -e1 ) };
	const cf s2 = pk_fma2( rs, rs, is * is );
	return cf{ __builtin_ldexpf( __builtin_amdgcn_sqrtf( s2.x ), e0 ), __builtin_ldexpf( __builtin_amdgcn_sqrtf( s2.y ), e1 ) };
	}
__device__ __forceinline__ void sincos_fast_2( cf x, cf & s, cf & c )
	{
	const cf kf = x * bcast2( 0x1.45f306p-1f );
	const cf k = cf{ __builtin_rintf( kf.x ), __builtin_rintf( kf.y ) };
	cf r = pk_fma2( -k, bcast2( 0x1.921fb6p+0f ), x );
	r = pk_fma2( -k, bcast2( -0x1.777a5cp-25f ), r );
	r = pk_fma2( -k, bcast2( -0x1.ee59dap-50f ), r );
	const cf r2 = r * r;
	cf sp = bcast2( 0x1.6cd1e4p-19f );
	sp = pk_fma2( sp, r2, bcast2( -0x1.a00f80p-13f ) );
	sp = pk_fma2( sp, r2, bcast2( 0x1.111108p-7f ) );
	sp = pk_fma2( sp, r2, bcast2( -0x1.555556p-3f ) );
	const cf sr = pk_fma2( r * r2, sp, r );
	cf cp = bcast2( 0x1.99eb7cp-16f );
	cp = pk_fma2( cp, r2, bcast2( -0x1.6c0c34p-10f ) );
	cp = pk_fma2( cp, r2, bcast2( 0x1.55554ap-5f ) );
	cp = pk_fma2( cp, r2, bcast2( -0x1.000000p-1f ) );
	const cf cr = pk_fma2( cp, r2, bcast2( 1.0f ) );
	const int q0 = int( k.x ), q1 = int( k.y );
	const float ss0 = ( q0 & 1 ) ? cr.x : sr.x, cc0 = ( q0 & 1 ) ? sr.x : cr.x;
	const float ss1 = ( q1 & 1 ) ? cr.y : sr.y, cc1 = ( q1 & 1 ) ? sr.y : cr.y;
	s = cf{ ( q0 & 2 ) ? -ss0 : ss0, ( q1 & 2 ) ? -ss1 : ss1 };
	c = cf{ ( ( q0 + 1 ) & 2 ) ? -cc0 : cc0, ( ( q1 + 1 ) & 2 ) ? -cc1 : cc1 };
	}

struct MFv { float m, f; };

// phase_vocoder.cpp:37-52.  prev_phase is the reference's phase_buffer (it only ever holds a float value).
template<bool FAST>
__device__ __forceinline__ MFv phase_vocode_bin_fast( float re, float im, float & prev_phase, float bin_frequency, float expected_phase_diff,
	float analysis_rate, bool use_wrapping )
	{
	const float phase = FAST ? atan2_fast( im, re ) : atan2f( im, re );               // std::arg
	const float phase_diff = float( double( phase ) - double( prev_phase ) );         // :44 double subtraction, narrowed
	prev_phase = phase;                                                               // :45
	const float delta_phase = phase_diff - expected_phase_diff;                       // :48
	const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * roundf( div_pi2( delta_phase ) ) : delta_phase; // :39-42,49
	const float delta_frequency = div_pi2( wrapped * analysis_rate );                 // :50
	MFv r;
	r.m = FAST ? magnitude_fast( re, im ) : hypotf( re, im );                         // std::abs
	r.f = bin_frequency + delta_frequency;                                            // :52
	return r;
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
	const float k = __builtin_rintf( x * 0x1.45f306p-1f );                            // 2/pi
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
	const int q = int( k );
	const float ss = ( q & 1 ) ? cr : sr;
	const float cc = ( q & 1 ) ? sr : cr;
	s = ( q & 2 ) ? -ss : ss;
	c = ( ( q + 1 ) & 2 ) ? -cc : cc;
	}

} // namespace flanhip
