// utility.hip -- mid/side conversion (Audio/AudioConversions.cpp:32-56) and the synthetic-input / comparison helpers
// used by bench.py and the tests (defined by this project).
#include "flanhip_internal.h"

namespace flanhip {

__global__ __launch_bounds__( 256 ) void k_mid_side( const float * in, int64_t n, float * out )
	{
	const int64_t i = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	if( i >= n ) return;
	const float sqrt2 = 1.41421356237309504880f;                  // std::sqrt( 2.0f )
	const float l = in[i], r = in[n + i];
	out[i]     = ( l + r ) / sqrt2;                               // AudioConversions.cpp:46
	out[n + i] = ( l - r ) / sqrt2;                               // :47
	}

// (16 bytes per store over the 16-byte aligned middle; the few floats in front of it and behind it one by one)
__global__ __launch_bounds__( 256 ) void k_fill( float * p, int64_t count, float v )
	{
	typedef float f4 __attribute__(( ext_vector_type( 4 ) ));
	const int64_t tid = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x, stride = int64_t( gridDim.x ) * blockDim.x;
	const int64_t head = min( count, int64_t( ( 16 - ( reinterpret_cast<uintptr_t>( p ) & 15 ) ) & 15 ) / 4 );   // floats before the first 16-byte boundary
	const int64_t quads = ( count - head ) / 4;
	f4 * q = reinterpret_cast<f4*>( p + head );
	const f4 vv = { v, v, v, v };
	for( int64_t i = tid; i < quads; i += stride ) q[i] = vv;
	const int64_t rest0 = head + 4 * quads;
	if( tid < head ) p[tid] = v;
	if( tid < count - rest0 ) p[rest0 + tid] = v;
	}

// The yardstick bench.py quotes the kernels against ("verify on the box with a copy kernel"): 16 bytes per lane and access, grid-stride,
// non-temporal both ways -- what a streaming kernel of this chip can move, read + written
__global__ __launch_bounds__( 256 ) void k_copy16( const float * __restrict__ src, float * __restrict__ dst, int64_t quads )
	{
	typedef float f4 __attribute__(( ext_vector_type( 4 ) ));
	const f4 * s = reinterpret_cast<const f4*>( src );
	f4 * d = reinterpret_cast<f4*>( dst );
	const int64_t stride = int64_t( gridDim.x ) * blockDim.x;
	int64_t i = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x;
	for( ; i + 3 * stride < quads; i += 4 * stride )
		{
		const f4 a = __builtin_nontemporal_load( s + i ), b = __builtin_nontemporal_load( s + i + stride );
		const f4 c = __builtin_nontemporal_load( s + i + 2 * stride ), e = __builtin_nontemporal_load( s + i + 3 * stride );
		__builtin_nontemporal_store( a, d + i ); __builtin_nontemporal_store( b, d + i + stride );
		__builtin_nontemporal_store( c, d + i + 2 * stride ); __builtin_nontemporal_store( e, d + i + 3 * stride );
		}
	for( ; i < quads; i += stride ) __builtin_nontemporal_store( __builtin_nontemporal_load( s + i ), d + i );
	}

__device__ __forceinline__ uint32_t hash32( uint32_t x )
	{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
	}

// uniform [-1,1) from a counter hash of (seed, channel, n): bit-identical to oracle_noise()
__global__ __launch_bounds__( 256 ) void k_noise( float * out, int64_t n, uint32_t seed )
	{
	const int c = blockIdx.y;
	const uint32_t base = hash32( seed ^ ( uint32_t( c ) * 0x9E3779B9U ) );
	for( int64_t i = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x; i < n; i += int64_t( gridDim.x ) * blockDim.x )
		{
		const uint32_t u = hash32( base + uint32_t( i ) * 0x85EBCA6BU + uint32_t( uint64_t( i ) >> 32 ) );
		out[int64_t( c ) * n + i] = float( u >> 8 ) * ( 1.0f / 8388608.0f ) - 1.0f;
		}
	}

__global__ __launch_bounds__( 256 ) void k_sqdiff( const float * a, const float * b, int64_t count, double * result )
	{
	double sd = 0.0, sb = 0.0;
	for( int64_t i = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x; i < count; i += int64_t( gridDim.x ) * blockDim.x )
		{
		const double x = a[i], y = b[i];
		sd += ( x - y ) * ( x - y ); sb += y * y;
		}
	for( int o = 32; o > 0; o >>= 1 ) { sd += __shfl_xor( sd, o ); sb += __shfl_xor( sb, o ); }
	if( ( threadIdx.x & 63 ) == 0 ) { atomicAdd( result, sd ); atomicAdd( result + 1, sb ); }
	}

} // namespace flanhip

using namespace flanhip;

extern "C" {

int flanhip_mid_side_dev( const float * d_in, int64_t n, float * d_out, void * stream )
	{
	FLANHIP_REQUIRE( d_in && d_out && n > 0, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	if( int rc = require_device() ) return rc;
	hipLaunchKernelGGL( k_mid_side, dim3( (unsigned) ( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, (hipStream_t) stream, d_in, n, d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_fill_dev( float * d_grid, int64_t count, float value, void * stream )
	{
	FLANHIP_REQUIRE( d_grid && count > 0, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	if( int rc = require_device() ) return rc;
	const unsigned blocks = (unsigned) std::min<int64_t>( ( count / 4 + 255 ) / 256 + 1, 4096 );
	hipLaunchKernelGGL( k_fill, dim3( blocks ), dim3( 256 ), 0, (hipStream_t) stream, d_grid, count, value );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_noise_dev( float * d_out, int64_t ch, int64_t n, uint32_t seed, void * stream )
	{
	FLANHIP_REQUIRE( d_out && ch > 0 && ch < 65536 && n > 0, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	if( int rc = require_device() ) return rc;
	const unsigned bx = (unsigned) std::min<int64_t>( ( n + 255 ) / 256, 2048 );
	hipLaunchKernelGGL( k_noise, dim3( bx, (unsigned) ch ), dim3( 256 ), 0, (hipStream_t) stream, d_out, n, seed );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_copy_dev( const float * d_src, float * d_dst, int64_t count, void * stream )
	{
	FLANHIP_REQUIRE( d_src && d_dst && count > 0 && count % 4 == 0, FLANHIP_ERR_INVALID_ARG, "bad arguments (a multiple of four floats)" );
	FLANHIP_REQUIRE( ( ( reinterpret_cast<uintptr_t>( d_src ) | reinterpret_cast<uintptr_t>( d_dst ) ) & 15 ) == 0, FLANHIP_ERR_INVALID_ARG, "16-byte aligned buffers" );
	if( int rc = require_device() ) return rc;
	const int64_t quads = count / 4;
	const unsigned blocks = (unsigned) std::min<int64_t>( ( quads + 255 ) / 256, int64_t( cu_count() ) * 16 );
	hipLaunchKernelGGL( k_copy16, dim3( blocks ), dim3( 256 ), 0, (hipStream_t) stream, d_src, d_dst, quads );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_sqdiff_dev( const float * d_a, const float * d_b, int64_t count, double * d_result, void * stream )
	{
	FLANHIP_REQUIRE( d_a && d_b && d_result && count > 0, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	if( int rc = require_device() ) return rc;
	FLANHIP_CHECK( hipMemsetAsync( d_result, 0, 2 * sizeof( double ), (hipStream_t) stream ) );
	const unsigned blocks = (unsigned) std::min<int64_t>( ( count + 255 ) / 256, 2048 );
	hipLaunchKernelGGL( k_sqdiff, dim3( blocks ), dim3( 256 ), 0, (hipStream_t) stream, d_a, d_b, count, d_result );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

} // extern "C"
