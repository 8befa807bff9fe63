// c_hooks.cpp -- plain-C doorway to the .flan PV file format of the host library (PVBuffer::save / load), so that other
// languages (and the parity tests, through ctypes) can read and write the files the reference exchanges
// (reference: PV/PVBuffer.cpp:99-140 save, :216-273 load).
#include <cstdint>
#include <cstring>
#include <limits>

#include "flan/PV.h"

extern "C" {

struct flan_pv_format { int32_t num_channels, num_frames, num_bins; float sample_rate, analysis_rate; int32_t window_size; };

int flan_pv_save_file( flan_pv_format f, const float * mf, const char * filename )
	{
	flan::PVBuffer::Format fmt;
	fmt.num_channels = f.num_channels; fmt.num_frames = f.num_frames; fmt.num_bins = f.num_bins;
	fmt.sample_rate = f.sample_rate; fmt.analysis_rate = f.analysis_rate; fmt.window_size = f.window_size;
	flan::PV pv = flan::PV::create_from_format( fmt );
	std::memcpy( pv.get_buffer().data(), mf, sizeof( flan::MF ) * pv.get_buffer().size() );
	return pv.save( filename ) ? 1 : 0;
	}

// returns the number of MFs in the file (0 on failure); copies them when they fit
int64_t flan_pv_load_file( const char * filename, flan_pv_format * f, float * mf, int64_t mf_capacity )
	{
	flan::PV pv = flan::PV::load_from_file( filename );
	const auto fmt = pv.get_format();
	f->num_channels = fmt.num_channels; f->num_frames = fmt.num_frames; f->num_bins = fmt.num_bins;
	f->sample_rate = fmt.sample_rate; f->analysis_rate = fmt.analysis_rate; f->window_size = fmt.window_size;
	const int64_t count = int64_t( pv.get_buffer().size() );
	if( mf && count <= mf_capacity ) std::memcpy( mf, pv.get_buffer().data(), sizeof( flan::MF ) * size_t( count ) );
	return count;
	}

// ---- Function<I,O>::sample of a few fixed callables: the doorway tests/test_ref_made_golden.py holds include/flan/Function.h to the grids the
// REFERENCE's own Function.h produced (tests/golden/ref_made/function_sample.npz; the callables are oracle/ref_driver.cpp's, index for index).
// which: 0  tf.t a + tf.f b    1  a step in t at a and in f at b    2  NaN where t f < 0, else t - f    3  the constant a
int64_t flan_function_sample2d( int which, float a, float b, int policy, float x0, float x1, float xs, float y0, float y1, float ys,
	float * out, int64_t capacity, int * is_constant, int64_t * small_dim )
	{
	using namespace flan;
	const ExecutionPolicy pol = static_cast<ExecutionPolicy>( policy );
	auto make = [&]() -> Function<TF, float>
		{
		switch( which )
			{
			case 0: return Function<TF, float>( [a, b]( TF tf ){ return tf.t * a + tf.f * b; }, pol );
			case 1: return Function<TF, float>( [a, b]( TF tf ){ return ( tf.t >= a ? 1.0f : 0.0f ) + ( tf.f >= b ? 2.0f : 0.0f ); }, pol );
			case 2: return Function<TF, float>( []( TF tf ){ return tf.t * tf.f < 0.0f ? std::numeric_limits<float>::quiet_NaN() : tf.t - tf.f; }, pol );
			default: return Function<TF, float>( a );
			}
		};
	const Function<TF, float> fn = make();
	const FunctionSample2d<float> s = fn.sample( x0, x1, xs, y0, y1, ys );
	*is_constant = s.is_constant() ? 1 : 0;
	*small_dim = int64_t( s.small_dim_size );
	if( s.is_constant() ) { if( capacity >= 1 ) out[0] = s.get_constant(); }
	else if( int64_t( s.size() ) <= capacity ) std::memcpy( out, s.get_vector().data(), sizeof( float ) * s.size() );
	return int64_t( s.size() );
	}

int64_t flan_function_sample1d( int which, float a, float b, int policy, int start, int end, float scale, float * out, int64_t capacity, int * is_constant )
	{
	using namespace flan;
	const ExecutionPolicy pol = static_cast<ExecutionPolicy>( policy );
	const Function<float, float> fn = which ? Function<float, float>( a ) : Function<float, float>( [a, b]( float x ){ return x * a + b; }, pol );
	const auto s = fn.sample( start, end, scale );
	*is_constant = s.is_constant() ? 1 : 0;
	if( s.is_constant() ) { if( capacity >= 1 ) out[0] = s.get_constant(); return int64_t( std::max( end - start, 0 ) ); }
	const int64_t n = int64_t( s.get_vector().size() );
	if( n <= capacity ) std::memcpy( out, s.get_vector().data(), sizeof( float ) * size_t( n ) );
	return n;
	}

} // extern "C"
