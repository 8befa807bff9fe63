// ref_driver.cpp -- extern "C" doorway into the REAL reference translation units that build here unmodified.
//
// TEST INFRASTRUCTURE ONLY.  Compiled (by oracle/Makefile, only where /root/reference exists) together with
// the reference's own sources, read in place from /root/reference/src:
//     flan/phase_vocoder.cpp  flan/WindowFunctions.cpp  flan/PV/PVBuffer.cpp  flan/Utility/Bytes.cpp
//     flan/Utility/buffer_access.cpp  flan/Utility/Interpolator.cpp  flan/Utility/{Interval,Color,Rect}.cpp
// and the header-only cubic spline the reference vendors (spline/spline.h, used by PV::stretch_spline, PV/PVModify.cpp:387-443)
// into oracle/_ref/libflanref.so.  No reference source is copied into this repository and no stand-in header or
// library is written: translation units that need FFTW3f / libsndfile / MSVC's std::_Pi (Conversions/AudioPV.cpp,
// PV/PVModify.cpp, PV/PV.cpp, Audio/*.cpp, FFTHelper.cpp) are simply NOT built -- see DESIGN.md.
//
// This file only forwards calls; it contains no algorithm.
#include <complex>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "flan/phase_vocoder.h"
#include "flan/WindowFunctions.h"
#include "flan/PV/PVBuffer.h"
#include "flan/Utility/Interpolator.h"
#include "spline/spline.h"

extern "C" {

void ref_phase_vocoder( double * phase_buffer, float re, float im, float bin_frequency, float analysis_rate, float sample_rate, float * m, float * f )
	{
	const flan::MF r = flan::phase_vocoder( *phase_buffer, std::complex<float>( re, im ), bin_frequency, analysis_rate, sample_rate );
	*m = r.m; *f = r.f;
	}

void ref_inverse_phase_vocoder( double * phase_buffer, float m, float f, float analysis_rate, float * re, float * im )
	{
	const std::complex<float> c = flan::inverse_phase_vocoder( *phase_buffer, flan::MF{ m, f }, analysis_rate );
	*re = c.real(); *im = c.imag();
	}

// Batched forms so that python can sweep millions of cases quickly.
void ref_phase_vocoder_batch( int64_t count, double * phase_buffer, const float * re, const float * im, const float * bin_frequency,
	float analysis_rate, float sample_rate, float * m, float * f )
	{
	for( int64_t i = 0; i < count; ++i )
		{
		const flan::MF r = flan::phase_vocoder( phase_buffer[i], std::complex<float>( re[i], im[i] ), bin_frequency[i], analysis_rate, sample_rate );
		m[i] = r.m; f[i] = r.f;
		}
	}

void ref_inverse_phase_vocoder_batch( int64_t count, double * phase_buffer, const float * m, const float * f, float analysis_rate, float * re, float * im )
	{
	for( int64_t i = 0; i < count; ++i )
		{
		const std::complex<float> c = flan::inverse_phase_vocoder( phase_buffer[i], flan::MF{ m[i], f[i] }, analysis_rate );
		re[i] = c.real(); im[i] = c.imag();
		}
	}

float ref_hann( float x ) { return flan::Windows::hann( x ); }

float ref_pi2() { return flan::pi2; }

// PVBuffer unit conversions on a buffer built with the given format.
struct RefPVFormat { int32_t num_channels, num_frames, num_bins; float sample_rate, analysis_rate; int32_t window_size; };

static flan::PVBuffer make_pv( const RefPVFormat & f )
	{
	flan::PVBuffer::Format fmt;
	fmt.num_channels = f.num_channels; fmt.num_frames = f.num_frames; fmt.num_bins = f.num_bins;
	fmt.sample_rate = f.sample_rate; fmt.analysis_rate = f.analysis_rate; fmt.window_size = f.window_size;
	return flan::PVBuffer( fmt );
	}

int   ref_pv_hop_size( RefPVFormat f )  { return make_pv( f ).get_hop_size(); }
int   ref_pv_dft_size( RefPVFormat f )  { return make_pv( f ).get_dft_size(); }
float ref_pv_bin_to_frequency( RefPVFormat f, float b ) { return make_pv( f ).bin_to_frequency( b ); }
float ref_pv_frequency_to_bin( RefPVFormat f, float x ) { return make_pv( f ).frequency_to_bin( x ); }
float ref_pv_time_to_frame( RefPVFormat f, float t )    { return make_pv( f ).time_to_frame( t ); }
float ref_pv_frame_to_time( RefPVFormat f, float x )    { return make_pv( f ).frame_to_time( x ); }
int64_t ref_pv_buffer_pos( RefPVFormat f, int c, int fr, int b ) { return (int64_t) make_pv( f ).get_buffer_pos( c, fr, b ); }
int ref_pv_is_nan_or_inf( RefPVFormat f, const float * mf )
	{
	flan::PVBuffer pv = make_pv( f );
	std::memcpy( pv.get_buffer().data(), mf, sizeof( flan::MF ) * pv.get_buffer().size() );
	return pv.is_nan_or_inf() ? 1 : 0;
	}

// .flan file format (PVBuffer.cpp:99-140 save, :216-273 load)
int ref_pv_save( RefPVFormat f, const float * mf, const char * filename )
	{
	flan::PVBuffer pv = make_pv( f );
	std::memcpy( pv.get_buffer().data(), mf, sizeof( flan::MF ) * pv.get_buffer().size() );
	return pv.save( filename ) ? 1 : 0;
	}

int ref_pv_load( const char * filename, RefPVFormat * f, float * mf, int64_t mf_capacity )
	{
	flan::PVBuffer pv( std::string{ filename } );
	const auto fmt = pv.get_format();
	f->num_channels = fmt.num_channels; f->num_frames = fmt.num_frames; f->num_bins = fmt.num_bins;
	f->sample_rate = fmt.sample_rate; f->analysis_rate = fmt.analysis_rate; f->window_size = fmt.window_size;
	const int64_t count = (int64_t) pv.get_buffer().size();
	if( mf && count <= mf_capacity )
		std::memcpy( mf, pv.get_buffer().data(), sizeof( flan::MF ) * count );
	return (int) ( count > 0 );
	}

// Utility/Interpolator.cpp: the named interpolators, numbered as oracle_interpolate numbers them
float ref_interpolate( int kind, float x )
	{
	using flan::Interpolator;
	switch( kind )
		{
		case 0: return Interpolator::linear()( x );
		case 1: return Interpolator::midpoint()( x );
		case 2: return Interpolator::nearest()( x );
		case 3: return Interpolator::floor()( x );
		case 4: return Interpolator::ceil()( x );
		case 5: return Interpolator::smoothstep()( x );
		case 6: return Interpolator::smootherstep()( x );
		case 7: return Interpolator::sqrt()( x );
		case 8: return Interpolator::sine()( x );
		}
	return x;
	}

// spline/spline.h as PV::stretch_spline uses it (PVModify.cpp:427-440): default boundary conditions, set_points( x, y ), then
// evaluated at nt points
void ref_spline( const double * x, const double * y, int n, const double * t, int nt, double * out )
	{
	tk::spline s;
	s.set_points( std::vector<double>( x, x + n ), std::vector<double>( y, y + n ) );
	for( int i = 0; i < nt; ++i ) out[i] = s( t[i] );
	}

} // extern "C"

// ---- Function<I,O>::sample / FunctionSample2d (flan/Function.h:141-171, FunctionSample.h:173-199): the reference's own templates, instantiated on
// a few fixed callables.  The header compiles unmodified here (its Graph / bitmap includes are declarations the driver never calls into).
#include "flan/Function.h"

namespace {
// which: 0  tf.t a + tf.f b      1  a step in t at a and in f at b      2  NaN where the product t f is negative, else t - f      3  the CONSTANT a
flan::Function<flan::TF, float> ref_fn2( int which, float a, float b, int policy )
	{
	const flan::ExecutionPolicy pol = static_cast<flan::ExecutionPolicy>( policy );
	switch( which )
		{
		case 0: return flan::Function<flan::TF, float>( [a, b]( flan::TF tf ){ return tf.t * a + tf.f * b; }, pol );
		case 1: return flan::Function<flan::TF, float>( [a, b]( flan::TF tf ){ return ( tf.t >= a ? 1.0f : 0.0f ) + ( tf.f >= b ? 2.0f : 0.0f ); }, pol );
		case 2: return flan::Function<flan::TF, float>( []( flan::TF tf ){ return tf.t * tf.f < 0.0f ? std::numeric_limits<float>::quiet_NaN() : tf.t - tf.f; }, pol );
		default: return flan::Function<flan::TF, float>( a );
		}
	}
}

extern "C" {

// the 2-D sample.  Returns vec_size (FunctionSample.h:21); *is_constant / *small_dim as the reference reports them; the grid (or the one constant) in out
// when it fits.  scan != 0: the in-place running sum down the frames of every bin through FunctionSample2d::at, exactly PV::stretch's loop
// (PV/PVModify.cpp:376-378) -- on a constant sample at() aliases ONE value, which is the doubling SURVEY 7 records.
int64_t ref_function_sample2d( int which, float a, float b, int policy, float x0, float x1, float xs, float y0, float y1, float ys, int scan,
	float * out, int64_t capacity, int * is_constant, int64_t * small_dim )
	{
	const flan::Function<flan::TF, float> fn = ref_fn2( which, a, b, policy );
	flan::FunctionSample2d<float> s = fn.sample( x0, x1, xs, y0, y1, ys );
	if( scan )
		{
		const int frames = int( s.size() / ( s.small_dim_size ? s.small_dim_size : 1 ) ), bins = int( s.small_dim_size );
		for( flan::Bin bin = 0; bin < bins; ++bin )
			for( flan::Frame frame = 1; frame < frames; ++frame )
				s.at( frame, bin ) += s.at( frame - 1, bin );
		}
	*is_constant = s.is_constant() ? 1 : 0;
	*small_dim = int64_t( s.small_dim_size );
	if( s.is_constant() ) { if( capacity >= 1 ) out[0] = s.get_constant(); }
	else if( int64_t( s.size() ) <= capacity ) std::memcpy( out, s.get_vector().data(), sizeof( float ) * s.size() );
	return int64_t( s.size() );
	}

// the 1-D sample (Function.h:141-153) of x a + b, or of the constant a (which != 0)
int64_t ref_function_sample1d( int which, float a, float b, int policy, int start, int end, float scale, float * out, int64_t capacity, int * is_constant )
	{
	const flan::ExecutionPolicy pol = static_cast<flan::ExecutionPolicy>( policy );
	const flan::Function<float, float> fn = which ? flan::Function<float, float>( a ) : flan::Function<float, float>( [a, b]( float x ){ return x * a + b; }, pol );
	flan::FunctionSample<float> s = fn.sample( start, end, scale );
	*is_constant = s.is_constant() ? 1 : 0;
	if( s.is_constant() ) { if( capacity >= 1 ) out[0] = s.get_constant(); }
	else if( int64_t( s.size() ) <= capacity ) std::memcpy( out, s.get_vector().data(), sizeof( float ) * s.size() );
	return int64_t( s.size() );
	}

} // extern "C" (Function)
