// processors_common.h -- small pieces shared by the frame-processor translation units (processors.hip, processors_ext.hip).
#pragma once
#include "flanhip_internal.h"
#include "fft_device.h"

namespace flanhip {

struct MFd { float m, f; };

// PVBuffer.cpp:428-431, :433-436, :438-441, :443-446
__device__ __forceinline__ float time_to_frame( float t, float sr, float hop ) { return t * sr / hop; }
__device__ __forceinline__ float frame_to_time( float f, float sr, float hop ) { return f / ( sr / hop ); }
__device__ __forceinline__ float frequency_to_bin( float f, float sr, float dft ) { return f / ( sr / dft ); }
__device__ __forceinline__ float bin_to_frequency( float b, float sr, float dft ) { return b * sr / dft; }

// float -> Bin / Frame.  The reference's plain conversion is undefined outside the int range (x86 yields INT_MIN there and
// for NaN); here it saturates and NaN becomes INT_MIN: every range check downstream rejects all of those values alike.
__device__ __forceinline__ int to_int_sat( float v )
	{
	return ( v == v ) ? int( v ) : INT_MIN;                                        // v_cvt_i32_f32 saturates
	}

// std::clamp( v, 0.0f, 1.0f ): NaN passes through
__device__ __forceinline__ float clamp01( float v ) { return v < 0.0f ? 0.0f : ( 1.0f < v ? 1.0f : v ); }

// Utility/Interpolator.cpp:14-101, numbered as include/flanhip.h numbers them (FLANHIP_INTERP_*)
__device__ __forceinline__ float interpolate( int kind, float x )
	{
	switch( kind )
		{
		case FLANHIP_INTERP_MIDPOINT:     return 0.5f;
		case FLANHIP_INTERP_NEAREST:      return roundf( x );
		case FLANHIP_INTERP_FLOOR:        return 0.0f;
		case FLANHIP_INTERP_CEIL:         return 1.0f;
		case FLANHIP_INTERP_SMOOTHSTEP:   return x * x * ( 3.0f - 2.0f * x );
		case FLANHIP_INTERP_SMOOTHERSTEP: return x * x * x * ( x * ( x * 6.0f - 15.0f ) + 10.0f );
		case FLANHIP_INTERP_SQRT:         return sqrtf( x );                                 // correctly rounded (hipcc default)
		case FLANHIP_INTERP_SINE:         return ( 1.0f - cosf( 3.14159274101257324f * x ) ) / 2.0f;   // cosf: within 1-2 ulp of libm's
		default:                          return x;                                                    // linear
		}
	}

// The "placement with conflicts" rule shared by PV::shape with shift alignment (PV.cpp:438-448) and PV::time_extrapolate
// (PVModify.cpp:658-664): candidates are visited in ascending source-bin order and one replaces the occupant of its target
// bin only if its magnitude is STRICTLY greater, the row starting from { 0, 0 }.  The survivor of a target bin is therefore
// the candidate with the greatest magnitude (> 0, not NaN), the lowest source bin among equals.  A wavefront resolves one
// row through LDS: key = ( magnitude bits << 32 | ~source bin ), ds_max_u64 per candidate, then the winners are written.
__device__ __forceinline__ void placement_offer( unsigned long long * keys, int target, float m, int source_bin )
	{
	if( m > 0.0f )                                                                 // beats the initial 0; false for NaN
		atomicMax( &keys[target], ( (unsigned long long) __float_as_uint( m ) << 32 ) | (unsigned long long) ( 0xFFFFFFFFu - unsigned( source_bin ) ) );
	}
__device__ __forceinline__ int placement_winner( unsigned long long key ) { return int( 0xFFFFFFFFu - unsigned( key & 0xFFFFFFFFull ) ); }

struct DevBuf
	{
	void * p = nullptr;
	~DevBuf() { if( p ) (void) hipFree( p ); }
	int alloc( size_t bytes ) { FLANHIP_CHECK( hipMalloc( &p, bytes ? bytes : 1 ) ); return FLANHIP_OK; }
	};

inline int check_pv_args( const void * a, const void * b, int64_t ch, int64_t F, int bins, float sr )
	{
	FLANHIP_REQUIRE( a && b, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	FLANHIP_REQUIRE( ch > 0 && F > 0 && bins >= 2 && sr > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad sizes" );
	return require_device();
	}

} // namespace flanhip
