// resample.hip -- Audio::resample for the 2:1 decimation case behind the C ABI (BASELINE config 5: 96 kHz -> 48 kHz).
//
// Reference: Audio/AudioConversions.cpp:14-30 calls r8b::CDSPResampler( src, dst, num_frames ) with default parameters
// (transition band 2 %, attenuation 206.91 dB, linear phase; r8brain/CDSPResampler.h:115-118) and ONE oneshot over the whole
// channel-major buffer -- all channels as a single stream.  For src = 2 dst that is one low-pass (CDSPFIRFilter::buildLPFilter,
// r8brain/CDSPFIRFilter.h:227-493, normalised cut-off 1/2, gain 1) applied by FFT block convolution with 2:1 decimation and the
// filter latency consumed (r8brain/CDSPBlockConvolver.h:62-184).  Convolution is convolution: here it is the direct fp64 sum
//        out[k] = float( sum_{j=-fl2..fl2} h[j] * x[2k - j] ),   x = 0 outside the buffer,
// with the 1621 taps of r8brain's Kaiser-power windowed sinc design computed once on the host.
// Other rate ratios run through r8brain's multi-stage interpolators and are not implemented: FLANHIP_ERR_UNSUPPORTED.
#include "flanhip_internal.h"
#include <cmath>
#include <vector>

namespace flanhip {

static double sqr( double x ) { return x * x; }

// Abramowitz & Stegun 9.8.1 / 9.8.2, the approximation of I0 r8brain windows with (r8bbase.h:1216-1236)
static double bessel_i0_as( double x )
	{
	const double ax = std::fabs( x );
	if( ax < 3.75 )
		{
		double y = x / 3.75; y *= y;
		return 1.0 + y * ( 3.5156229 + y * ( 3.0899424 + y * ( 1.2067492 + y * ( 0.2659732 + y * ( 0.360768e-1 + y * 0.45813e-2 ) ) ) ) );
		}
	const double y = 3.75 / ax;
	return std::exp( ax ) / std::sqrt( ax ) * ( 0.39894228 + y * ( 0.1328592e-1 + y * ( 0.225319e-2 + y * ( -0.157565e-2 + y * ( 0.916281e-2 +
		y * ( -0.2057706e-1 + y * ( 0.2635537e-1 + y * ( -0.1647633e-1 + y * 0.392377e-2 ) ) ) ) ) ) ) );
	}

// buildLPFilter for the parameters Flan always uses (tb 2 %, 206.91 dB, linear phase), cut-off `req_norm_freq`, DC gain `gain`
static bool design_default_lowpass( double req_norm_freq, double gain, std::vector<double> & h, int & fl2 )
	{
	const double tb = 2.0 * 0.01;                                              // CDSPFIRFilter.h:229
	double atten = -206.91 - 0.21;                                             // :233, :268-273
	const int corr_index = int( std::floor( ( -atten - 49.0 ) * 264 / 176.25 + 0.5 ) );   // :285-291
	if( corr_index != 237 ) return false;
	atten -= -19 / 196.0;                                                      // :354-376, entry 237 of the tb < 0.10 correction table
	const double pwr = 7.43932822146293e-8 * sqr( atten ) + 0.000102747434588003 * std::cos( 0.00785021930010397 * atten ) *
		std::cos( 0.633854318781239 + 0.103208573657699 * atten ) - 0.00798132247867036 - 0.000903555213543865 * atten -
		0.0969365532127236 * std::exp( 0.0779275237937911 * atten ) - 1.37304948662012e-5 * atten * std::cos( 0.00785021930010397 * atten );   // :379-384
	double hl, fo1;
	if( pwr <= 0.067665322581 )                                                // :386, :425-435
		{
		hl = ( 2.45739657014937 + 269.183679500541 * pwr * std::cos( 5.73225668178813 +
			std::atan2( std::cosh( 0.988861169868941 - 17.2201556280744 * pwr ), 1.08340138240431 * pwr ) ) ) / tb;
		fo1 = 2.291956939 * tb + 0.01942450693 * sqr( tb ) * hl - 4.67538973161837 * pwr * tb - 1.668433124 * tb * std::pow( pwr, pwr );
		}
	else                                                                       // :450-461
		{
		hl = ( 1.15990238966306 * pwr - 5.02124037125213 * sqr( pwr ) - 0.158676856669827 * atten *
			std::cos( 1.1609073390614 * pwr - 6.33932586197475 * pwr * sqr( pwr ) ) ) / tb;
		fo1 = 0.867344453126885 * tb + 0.052693817907757 * tb * std::log( pwr ) + 0.0895511178735932 * tb * std::atan( 59.7538527741309 * pwr ) -
			0.0745653568081453 * pwr * tb;
		}
	const double pi = 3.14159265358979323846;
	const double len2 = 0.25 * hl / req_norm_freq;                             // :468
	const double freq2 = pi * ( 1.0 - fo1 ) * req_norm_freq;                   // :470
	fl2 = int( std::floor( len2 ) );                                           // CDSPSincFilterGen.h:136
	const double beta = 125.0, kdiv = bessel_i0_as( beta );                    // CDSPFIRFilter.h:464, CDSPSincFilterGen.h:584-588
	h.assign( 2 * fl2 + 1, 0.0 );
	auto window = [&]( int t )                                                 // CDSPSincFilterGen.h:246-257, to the power pwr (:354)
		{
		const double n = 1.0 - sqr( t / len2 );
		return std::pow( n < 0.0 ? 0.0 : bessel_i0_as( beta * std::sqrt( n ) ) / kdiv, pwr );
		};
	h[fl2] = freq2 * window( 0 ) / pi;                                         // :353
	for( int t = 1; t <= fl2; ++t )                                            // :356-365
		{
		const double v = std::sin( freq2 * t ) * window( t ) / t / pi;
		h[fl2 + t] = v; h[fl2 - t] = v;
		}
	double s = 0.0;                                                            // CDSPFIRFilter.h:496-507
	for( double v : h ) s += v;
	for( double & v : h ) v *= gain / s;
	return true;
	}

struct ResamplePlan { double * d_taps = nullptr; int fl2 = 0; };
static std::mutex g_rs_mutex;
static std::map<int, ResamplePlan> g_rs_plans;   // per device

static int get_resample_plan( const ResamplePlan ** out )
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	std::lock_guard<std::mutex> lock( g_rs_mutex );
	auto it = g_rs_plans.find( device );
	if( it != g_rs_plans.end() ) { *out = &it->second; return FLANHIP_OK; }
	std::vector<double> h; int fl2 = 0;
	FLANHIP_REQUIRE( design_default_lowpass( 0.5, 1.0, h, fl2 ), FLANHIP_ERR_UNSUPPORTED, "low-pass design outside the restated range" );
	ResamplePlan p; p.fl2 = fl2;
	FLANHIP_CHECK( hipMalloc( &p.d_taps, sizeof( double ) * h.size() ) );
	FLANHIP_CHECK( hipMemcpy( p.d_taps, h.data(), sizeof( double ) * h.size(), hipMemcpyHostToDevice ) );
	*out = &g_rs_plans.emplace( device, p ).first->second;
	return FLANHIP_OK;
	}

// One block = 256 consecutive outputs of the stream.  The 512 + 2*fl2 input samples the block needs and the taps are staged in
// LDS (fp64); each thread runs the 2*fl2+1-term sum with fma, input index ascending.
constexpr int RS_BLOCK = 256;
__global__ __launch_bounds__( RS_BLOCK ) void k_resample_2to1( const float * in, int64_t total_in, const double * taps, int fl2, float * out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int ntaps = 2 * fl2 + 1;
	double * s_h = reinterpret_cast<double*>( smem );                       // [ntaps]
	double * s_x = s_h + ntaps + 1;                                         // [2*RS_BLOCK + 2*fl2]
	const int64_t k0 = int64_t( blockIdx.x ) * RS_BLOCK;
	const int64_t x0 = 2 * k0 - fl2;                                        // first input sample the block touches
	const int span = 2 * RS_BLOCK + 2 * fl2;
	for( int i = threadIdx.x; i < ntaps; i += RS_BLOCK ) s_h[i] = taps[i];
	for( int i = threadIdx.x; i < span; i += RS_BLOCK )
		{
		const int64_t a = x0 + i;
		s_x[i] = ( a >= 0 && a < total_in ) ? double( in[a] ) : 0.0;
		}
	__syncthreads();
	const int64_t k = k0 + threadIdx.x;
	if( k >= total_out ) return;
	// out[k] = sum_i h[fl2 + (2k - i)] x[i], i = 2k-fl2 .. 2k+fl2  ->  local index i - x0 = 2*tid + m, tap index ntaps-1-m
	const double * xp = s_x + 2 * threadIdx.x;
	double acc = 0.0;
	#pragma unroll 4
	for( int m = 0; m < ntaps; ++m ) acc = __builtin_fma( s_h[ntaps - 1 - m], xp[m], acc );
	out[k] = float( acc );
	}

} // namespace flanhip

using namespace flanhip;

extern "C" {

int64_t flanhip_resample_out_frames( int64_t num_frames, float src_rate, float dst_rate )
	{
	// AudioConversions.cpp:22  format.num_frames *= new_sample_rate / get_sample_rate()   (Frame *= float)
	return int64_t( int32_t( float( int32_t( num_frames ) ) * ( dst_rate / src_rate ) ) );
	}

int flanhip_resample_dev( const float * d_in, int64_t ch, int64_t n, float src_rate, float dst_rate, float * d_out, void * stream )
	{
	FLANHIP_REQUIRE( d_in && d_out && ch > 0 && n > 0 && src_rate > 0.0f && dst_rate > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	FLANHIP_REQUIRE( double( src_rate ) == 2.0 * double( dst_rate ), FLANHIP_ERR_UNSUPPORTED, "only the 2:1 (e.g. 96 kHz -> 48 kHz) ratio is implemented" );
	if( int rc = require_device() ) return rc;
	const ResamplePlan * plan = nullptr;
	if( int rc = get_resample_plan( &plan ) ) return rc;
	const int64_t n_out = flanhip_resample_out_frames( n, src_rate, dst_rate );
	const int64_t total_in = ch * n, total_out = ch * n_out;
	if( total_out <= 0 ) return FLANHIP_OK;
	const size_t lds = sizeof( double ) * ( size_t( 2 * plan->fl2 + 2 ) + size_t( 2 * RS_BLOCK + 2 * plan->fl2 ) );
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_resample_2to1 ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	hipLaunchKernelGGL( k_resample_2to1, dim3( (unsigned) ( ( total_out + RS_BLOCK - 1 ) / RS_BLOCK ) ), dim3( RS_BLOCK ), lds, (hipStream_t) stream,
		d_in, total_in, plan->d_taps, plan->fl2, d_out, total_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_resample( const float * in, int64_t ch, int64_t n, float src_rate, float dst_rate, float * out, volatile int * cancel )
	{
	FLANHIP_REQUIRE( in && out && ch > 0 && n > 0, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	FLANHIP_REQUIRE( double( src_rate ) == 2.0 * double( dst_rate ), FLANHIP_ERR_UNSUPPORTED, "only the 2:1 (e.g. 96 kHz -> 48 kHz) ratio is implemented" );
	if( int rc = require_device() ) return rc;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	const int64_t n_out = flanhip_resample_out_frames( n, src_rate, dst_rate );
	float * d_in = nullptr; float * d_out = nullptr;
	FLANHIP_CHECK( hipMalloc( &d_in, sizeof( float ) * size_t( ch * n ) ) );
	if( hipMalloc( &d_out, sizeof( float ) * size_t( std::max<int64_t>( ch * n_out, 1 ) ) ) != hipSuccess ) { (void) hipFree( d_in ); set_error( "hipMalloc failed" ); return FLANHIP_ERR_HIP; }
	int rc = FLANHIP_OK;
	if( hipMemcpy( d_in, in, sizeof( float ) * size_t( ch * n ), hipMemcpyHostToDevice ) != hipSuccess ) { set_error( "upload failed" ); rc = FLANHIP_ERR_HIP; }
	if( !rc ) rc = flanhip_resample_dev( d_in, ch, n, src_rate, dst_rate, d_out, nullptr );
	if( !rc && hipDeviceSynchronize() != hipSuccess ) { set_error( "resample kernel failed" ); rc = FLANHIP_ERR_HIP; }
	if( !rc && cancelled( cancel ) ) rc = FLANHIP_ERR_CANCELLED;
	if( !rc && hipMemcpy( out, d_out, sizeof( float ) * size_t( ch * n_out ), hipMemcpyDeviceToHost ) != hipSuccess ) { set_error( "download failed" ); rc = FLANHIP_ERR_HIP; }
	(void) hipFree( d_in ); (void) hipFree( d_out );
	return rc;
	}

} // extern "C"
