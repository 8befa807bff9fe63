// c_hooks.cpp -- plain-C doorway to the .flan PV file format of the host library (PVBuffer::save / load), so that other
// languages (and the parity tests, through ctypes) can read and write the files the reference exchanges
// (reference: PV/PVBuffer.cpp:99-140 save, :216-273 load).
#include <cstdint>
#include <cstring>

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

} // extern "C"
