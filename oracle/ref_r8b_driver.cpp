// ref_r8b_driver.cpp -- extern "C" doorway into the REAL r8brain-free-src resampler vendored by the reference
// (/root/reference/src/r8brain, header-only + r8bbase.cpp; builds here unmodified with g++, no external library).
// TEST INFRASTRUCTURE ONLY; forwards calls, contains no algorithm.  Mirrors exactly what Audio::resample does with it
// (Audio/AudioConversions.cpp:25-27): one CDSPResampler( src, dst, MaxInLen ) with default parameters, one
// oneshot<float,float> over the WHOLE channel-major buffer.
#include "r8brain/CDSPResampler.h"

extern "C" {

int ref_r8b_resample( const float * in, int in_len, double src_rate, double dst_rate, int max_in_len, float * out, int out_len )
	{
	r8b::CDSPResampler resampler( src_rate, dst_rate, max_in_len );
	resampler.oneshot<float, float>( in, in_len, out, out_len );
	return 0;
	}

} // extern "C"
