// pv_kernels_any.h -- Audio::convert_to_PV / PV::convert_to_audio for ANY even dft size.
//
// The reference hands whatever dft_size the caller passes to FFTW (FFTHelper.cpp:16-26, called from Conversions/AudioPV.cpp:38,106;
// Audio.h:158-163 puts no constraint on it): convert_to_PV( 2048, 512, 3000 ) or ( ..., 16384 ) are ordinary calls.  The tuned and the
// LDS-resident kernels serve powers of two up to 8192; everything else -- other even sizes, larger powers of two -- comes here.
// Correctness first, speed second: the transform is the DEFINITION, a direct sum in fp64
//     X[k] = sum_n s[n] exp( -2 pi i k n / N ),   n < window_size (the rest of the frame is zero padding, AudioPV.cpp:65)
// one thread per bin, the phasor advanced by a complex multiplication per sample and re-seeded from an exact table every 64 samples
// (error ~1e-14: the result rounds to fp32 like an exact transform's), FOUR frames per pass over the samples so that the phasor's
// cost is shared.  O( window x bins ) per frame: 8 ch x 60 s at ( 2048, 512, 3000 ) takes milliseconds, not the fraction of one the
// power-of-two kernels need -- and nothing is refused.  The per-bin phase-vocoder arithmetic is the generic kernels' (pv_kernels.h:
// phase_vocode_bin and the inverse), rounding for rounding.
// Synthesis: k_any_spectra (inverse phase vocoder along chains of frames, from the carries of the common pre-pass) -> k_any_inverse (the
// c2r definition, x[n] = X0 + (-1)^n X[N/2] + 2 sum Re( X[k] e^{+2 pi i k n / N} ), n < window_size, windowed) -> k_any_overlap_add (each
// output sample sums its frames in ascending order: the reference's own order, AudioPV.cpp:113-135, no atomics).
#pragma once
#include "pv_kernels.h"

namespace flanhip {

constexpr int ANY_FB = 4;            // frames per pass
constexpr int ANY_TILE = 256;        // samples (analysis) / bins (synthesis) staged per step
constexpr int ANY_THREADS = 256;
constexpr int ANY_RESEED = 64;

// unit[m] = ( cos( 2 pi m / N ), sin( 2 pi m / N ) ), exact at the quarter turns
__device__ __forceinline__ d2 any_phasor( const d2 * __restrict__ unit, int N, int64_t a, int64_t b, bool forward )
	{
	const d2 u = unit[int( ( a * b ) % N )];
	return d2{ u.x, forward ? -u.y : u.y };
	}

// ---- analysis: grid ( ceil( bins / 256 ), channels * chains_per_channel ) -----------------------------------------------------------
__global__ __launch_bounds__( ANY_THREADS ) void k_analyze_any( AnalyzeParams p, const d2 * __restrict__ unit, int N )
	{
	__shared__ double s_tile[ANY_FB][ANY_TILE];
	const int tid = threadIdx.x;
	const int bins = N / 2 + 1;
	const int k = int( blockIdx.x ) * ANY_THREADS + tid;
	const bool have_bin = k < bins;
	const int channel = int( blockIdx.y ) / p.chains_per_channel, chain = int( blockIdx.y ) % p.chains_per_channel;
	const int64_t t0 = int64_t( chain ) * p.L, t1 = min( t0 + int64_t( p.L ), p.F );
	const int64_t tfirst = t0 > 0 ? t0 - 1 : t0;                                // the halo frame lends its phases (phase_vocoder.cpp:45)
	const float * x = p.audio + int64_t( channel ) * p.n;
	const int W = p.window_size, hop = p.hop;
	const bool use_wrapping = p.analysis_rate < p.sample_rate;                    // phase_vocoder.cpp:37
	const float binf = float( k ) * p.sample_rate / float( N );                   // PVBuffer.cpp:443-446
	const float expected = binf / p.analysis_rate * FLANHIP_PI2_F;                // phase_vocoder.cpp:47
	const d2 step = have_bin ? d2{ unit[k].x, -unit[k].y } : d2{ 1.0, 0.0 };       // exp( -2 pi i k / N )
	float prev = 0.0f;                                                            // AudioPV.cpp:44
	// cancellation (core.hip): these launches can last tenths of a second, so besides the block's start the word is read again every 8 batches
	// of frames -- by thread 0, published through LDS at the barriers the sample tiles need anyway
	__shared__ int s_cancel;
	int batch = 0;
	for( int64_t tb = tfirst; tb < t1; tb += ANY_FB, ++batch )
		{
		d2 acc[ANY_FB];
		#pragma unroll
		for( int f = 0; f < ANY_FB; ++f ) acc[f] = d2{ 0.0, 0.0 };
		for( int n0 = 0; n0 < W; n0 += ANY_TILE )
			{
			__syncthreads();
			if( tid == 0 && n0 == 0 && ( batch & 7 ) == 0 ) s_cancel = cancel_peek( p.cancel );   // written between the two barriers, read after the second
			#pragma unroll
			for( int f = 0; f < ANY_FB; ++f )
				{
				const int n = n0 + tid;
				const int64_t t = tb + f;
				const int64_t a = int64_t( hop ) * t - W / 2 + n;                   // AudioPV.cpp:52
				float v = 0.0f;
				if( n < W && t < t1 && a >= 0 && a < p.n ) v = x[a] * p.window[n];  // :54-62 (a float product)
				s_tile[f][tid] = double( v );
				}
			__syncthreads();
			if( s_cancel ) return;                                                  // (block-uniform: everybody reads it after the same barrier)
			if( have_bin )
				{
				const int len = min( ANY_TILE, W - n0 );
				d2 z = d2{ 1.0, 0.0 };
				for( int i = 0; i < len; ++i )
					{
					if( ( i & ( ANY_RESEED - 1 ) ) == 0 ) z = any_phasor( unit, N, k, n0 + i, true );
					#pragma unroll
					for( int f = 0; f < ANY_FB; ++f )
						{
						const double sv = s_tile[f][i];
						acc[f].x = __builtin_fma( sv, z.x, acc[f].x );
						acc[f].y = __builtin_fma( sv, z.y, acc[f].y );
						}
					z = d2{ __builtin_fma( z.x, step.x, -( z.y * step.y ) ), __builtin_fma( z.x, step.y, z.y * step.x ) };
					}
				}
			}
		if( have_bin )
			{
			#pragma unroll
			for( int f = 0; f < ANY_FB; ++f )
				{
				const int64_t t = tb + f;
				if( t < t1 )
					{
					const float re = float( acc[f].x );
					const float im = ( k == 0 || k == N / 2 ) ? 0.0f : float( acc[f].y );   // r2c: X[0], X[N/2] are real
					const MF mf = phase_vocode_bin( re, im, prev, binf, expected, p.analysis_rate, use_wrapping );
					if( t >= t0 ) p.out[( int64_t( channel ) * p.F + t ) * bins + k] = mf;
					}
				}
			}
		}
	}

// ---- synthesis ----------------------------------------------------------------------------------------------------------------------
struct AnySynthParams
	{
	const MF * pv; const double * carry; float * spec; float * frames; float * out; const float * window;
	int64_t F, out_len; int num_channels, bins, N, W, hop, L, chains_per_channel; float analysis_rate, window_scale;
	const int * cancel;       // the launching thread's cancel word (core.hip), or null
	};

// inverse phase vocoder (AudioPV.cpp:117-120, phase_vocoder.cpp:55-61): thread = ( chain, bin ), frames in order; spec[ch][F][bins] = X
__global__ __launch_bounds__( ANY_THREADS ) void k_any_spectra( AnySynthParams p )
	{
	const int k = int( blockIdx.x ) * ANY_THREADS + threadIdx.x;
	if( k >= p.bins || cancel_seen( cancel_peek( p.cancel ) ) ) return;          // (no barriers in this kernel)
	const int channel = int( blockIdx.y ) / p.chains_per_channel, chain = int( blockIdx.y ) % p.chains_per_channel;
	const int64_t t0 = int64_t( chain ) * p.L, t1 = min( t0 + int64_t( p.L ), p.F );
	double phase = p.carry[( int64_t( channel ) * p.chains_per_channel + chain ) * p.bins + k];
	for( int64_t t = t0; t < t1; ++t )
		{
		const int64_t at = ( int64_t( channel ) * p.F + t ) * p.bins + k;
		const MF mf = p.pv[at];
		phase += double( phase_term( mf.f, p.analysis_rate ) );                  // :57-58
		phase = ( __builtin_fabs( phase ) < FLANHIP_FOLD_FAST_LIMIT ) ? fold_phase_fast( phase ) : fold_phase_any( phase );   // :59
		const float th = float( phase );
		float sn, cs;
		if( __builtin_fabsf( th ) < FLANHIP_SINCOS_FAST_LIMIT ) sincos_fast( th, sn, cs );
		else { const float2 sc = sincos_wide( th ); sn = sc.x; cs = sc.y; }
		reinterpret_cast<cf*>( p.spec )[at] = mk( mf.m * cs, mf.m * sn );         // std::polar, :60
		}
	}

// c2r by its definition (unnormalised; Im X[0] and Im X[N/2] ignored, like fftwf_plan_dft_c2r_1d), then the window (AudioPV.cpp:133-134):
// grid ( ceil( W / 256 ), ceil( channels * F / 4 ) ), thread = sample n of four frames; frames[ch][F][W]
__global__ __launch_bounds__( ANY_THREADS ) void k_any_inverse( AnySynthParams p, const d2 * __restrict__ unit )
	{
	__shared__ d2 s_spec[ANY_FB][ANY_TILE];
	const int tid = threadIdx.x;
	const int n = int( blockIdx.x ) * ANY_THREADS + tid;
	const bool have = n < p.W;
	const int64_t row0 = int64_t( blockIdx.y ) * ANY_FB, rows = int64_t( p.num_channels ) * p.F;
	const int N = p.N, half = N / 2;
	const d2 step = have ? unit[n] : d2{ 1.0, 0.0 };                              // exp( +2 pi i n / N )
	double acc[ANY_FB];
	#pragma unroll
	for( int f = 0; f < ANY_FB; ++f ) acc[f] = 0.0;
	__shared__ int s_cancel;
	if( tid == 0 ) s_cancel = cancel_peek( p.cancel );
	__syncthreads();
	if( s_cancel ) return;
	for( int k0 = 0; k0 < p.bins; k0 += ANY_TILE )
		{
		__syncthreads();
		#pragma unroll
		for( int f = 0; f < ANY_FB; ++f )
			{
			const int k = k0 + tid;
			d2 v = d2{ 0.0, 0.0 };
			if( k < p.bins && row0 + f < rows )
				{
				const cf xk = reinterpret_cast<const cf*>( p.spec )[( row0 + f ) * p.bins + k];
				const bool edge = k == 0 || k == half;
				v = d2{ edge ? double( xk.x ) : 2.0 * double( xk.x ), edge ? 0.0 : 2.0 * double( xk.y ) };
				}
			s_spec[f][tid] = v;
			}
		__syncthreads();
		if( have )
			{
			const int len = min( ANY_TILE, p.bins - k0 );
			d2 z = d2{ 1.0, 0.0 };
			for( int i = 0; i < len; ++i )
				{
				if( ( i & ( ANY_RESEED - 1 ) ) == 0 ) z = any_phasor( unit, N, n, k0 + i, false );
				#pragma unroll
				for( int f = 0; f < ANY_FB; ++f )
					{
					const d2 v = s_spec[f][i];
					acc[f] = __builtin_fma( v.x, z.x, __builtin_fma( -v.y, z.y, acc[f] ) );     // Re( X e^{ i theta } )
					}
				z = d2{ __builtin_fma( z.x, step.x, -( z.y * step.y ) ), __builtin_fma( z.x, step.y, z.y * step.x ) };
				}
			}
		}
	if( have )
		{
		const float w = p.window[n] * p.window_scale;                             // AudioPV.cpp:102
		#pragma unroll
		for( int f = 0; f < ANY_FB; ++f )
			if( row0 + f < rows ) p.frames[( row0 + f ) * p.W + n] = float( acc[f] ) * w;   // :134's product
		}
	}

// out[ch][F hop]: every sample adds the frames that cover it in ascending frame order onto the zero it starts as (AudioPV.cpp:113-135)
__global__ __launch_bounds__( 256 ) void k_any_overlap_add( AnySynthParams p )
	{
	const int64_t idx = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( idx >= int64_t( p.num_channels ) * p.out_len ) return;
	const int channel = int( idx / p.out_len );
	const int64_t pos = idx % p.out_len;
	// frames t with 0 <= pos - ( hop t - W/2 ) < W
	const int64_t hi = ( pos + p.W / 2 ) / p.hop;                                // (the numerator is >= 0)
	const int64_t num = pos + p.W / 2 - p.W;                                     // hop t > num
	int64_t lo = num < 0 ? 0 : num / p.hop + 1;
	const int64_t last = min( hi, p.F - 1 );
	float acc = 0.0f;
	for( int64_t t = lo; t <= last; ++t )
		acc += p.frames[( int64_t( channel ) * p.F + t ) * p.W + ( pos - ( int64_t( p.hop ) * t - p.W / 2 ) )];
	p.out[idx] = acc;
	}

} // namespace flanhip
