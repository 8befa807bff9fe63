// pv_math.h -- per-bin arithmetic of the phase vocoder with the reference's rounding sequence
// (phase_vocoder.cpp:37-61), written so that the expensive steps cost a few instructions on gfx950.
//
// Everything here is compiled with -ffp-contract=off: a*b+c is two roundings unless fmaf() is written out.
#pragma once
#include <hip/hip_runtime.h>

namespace flanhip {

// defines.h:44-45: pi = acos(-1.0f) (float), pi2 = pi * 2.0f  -> 6.2831854820251465 as a float, NOT 2*pi.
#define FLANHIP_PI2_F 6.2831854820251465f
#define FLANHIP_PI2_D 6.2831854820251465
#define FLANHIP_RPI2_F 0x1.45f306p-3f       /* RN( 1 / pi2 ) */

// x / pi2 with IEEE round-to-nearest semantics in 3 instructions.  q0 = RN(x*rc); r = x - q0*c exactly (fma);
// q = RN(q0 + r*rc) is the correctly rounded quotient for EVERY float with |x| >= 1e-30 -- checked exhaustively
// over all 3.8e9 such floats by tools/check_div_pi2.cpp; smaller |x| (the residual would be subnormal) take the
// hardware division.
__device__ __forceinline__ float div_pi2( float x )
	{
	if( __builtin_fabsf( x ) < 1.0e-30f ) return x / FLANHIP_PI2_F;
	const float q0 = x * FLANHIP_RPI2_F;
	const float r = __builtin_fmaf( -q0, FLANHIP_PI2_F, x );
	return __builtin_fmaf( r, FLANHIP_RPI2_F, q0 );
	}

// |z| for spectra of audio-scale signals: one fma, one correctly rounded sqrt (within 1 ulp of hypotf); the scaled
// path keeps the result finite/accurate when the squares would overflow or underflow.
__device__ __forceinline__ float magnitude( float re, float im )
	{
	const float a = __builtin_fmaxf( __builtin_fabsf( re ), __builtin_fabsf( im ) );
	if( a > 1.0e18f || ( a < 1.0e-18f && a > 0.0f ) ) return hypotf( re, im );
	return __builtin_sqrtf( __builtin_fmaf( re, re, im * im ) );
	}

struct MFv { float m, f; };

// phase_vocoder.cpp:37-52.  prev_phase is the reference's phase_buffer (it only ever holds a float value).
__device__ __forceinline__ MFv phase_vocode_bin_fast( float re, float im, float & prev_phase, float bin_frequency, float expected_phase_diff,
	float analysis_rate, bool use_wrapping )
	{
	const float phase = atan2f( im, re );                                             // std::arg
	const float phase_diff = float( double( phase ) - double( prev_phase ) );         // :44 double subtraction, narrowed
	prev_phase = phase;                                                               // :45
	const float delta_phase = phase_diff - expected_phase_diff;                       // :48
	const float wrapped = use_wrapping ? delta_phase - FLANHIP_PI2_F * roundf( div_pi2( delta_phase ) ) : delta_phase; // :39-42,49
	const float delta_frequency = div_pi2( wrapped * analysis_rate );                 // :50
	MFv r;
	r.m = magnitude( re, im );                                                        // std::abs
	r.f = bin_frequency + delta_frequency;                                            // :52
	return r;
	}

// phase_vocoder.cpp:57-59:  phase_buffer += term; if( phase_buffer > pi2 ) phase_buffer = fmod( phase_buffer, pi2 ).
// fmod is exact; for ph < 1e6 the quotient is < 2^18, q*pi2 (24-bit constant) is exact in double and so is the
// difference, so floor + fma + one correction reproduces it without a division.
__device__ __forceinline__ double fold_phase_fast( double ph )
	{
	if( ph > FLANHIP_PI2_D )
		{
		if( ph < 1.0e6 )
			{
			const double q = __builtin_floor( ph * ( 1.0 / FLANHIP_PI2_D ) );
			double r = __builtin_fma( -q, FLANHIP_PI2_D, ph );
			if( r < 0.0 ) r += FLANHIP_PI2_D;
			else if( r >= FLANHIP_PI2_D ) r -= FLANHIP_PI2_D;
			ph = r;
			}
		else ph = fmod( ph, FLANHIP_PI2_D );
		}
	return ph;
	}

// sin and cos of a float angle, ~1 ulp, for the range the folded phase lives in; anything larger goes to sincosf.
__device__ __forceinline__ void sincos_fast( float x, float & s, float & c )
	{
	if( !( __builtin_fabsf( x ) < 8192.0f ) ) { sincosf( x, &s, &c ); return; }
	// Cody-Waite: r = x - k*pi/2 with pi/2 split in three (fma keeps every partial product exact enough for k < 2^13)
	const float k = __builtin_rintf( x * 0x1.45f306p-1f );                            // 2/pi
	float r = __builtin_fmaf( -k, 0x1.921fb6p+0f, x );
	r = __builtin_fmaf( -k, -0x1.777a5cp-25f, r );
	r = __builtin_fmaf( -k, -0x1.ee59dap-50f, r );
	const float r2 = r * r;
	// minimax fits on [-pi/4, pi/4] (tools/fit_sincos.py): max abs error 7e-8 over [-8192, 8192]
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
