// AudioBuffer.cpp -- host/device mirrored audio container (reference: src/flan/Audio/AudioBuffer.cpp:17-29,479-482).
#include "flan/AudioBuffer.h"

#include <algorithm>
#include <utility>

#include <iostream>

#include <cmath>
#include "device_block.h"

namespace flan {

AudioBuffer::AudioBuffer() : format(), buffer() {}

AudioBuffer::AudioBuffer( const Format & other ) : format( other ), buffer( count() ) {}

AudioBuffer::AudioBuffer( std::vector<float> && temp_buffer, Channel num_channels, FrameRate sr )
	: format(), buffer( std::move( temp_buffer ) )
	{
	format.num_channels = num_channels;
	format.num_frames = num_channels > 0 ? Frame( buffer.size() / num_channels ) : 0;   // AudioBuffer.cpp:22
	format.sample_rate = sr;
	}

AudioBuffer AudioBuffer::adopt_device( const Format & f, std::shared_ptr<detail::DeviceBlock> block )
	{
	AudioBuffer out;
	out.format = f;
	out.dev = std::move( block );
	out.host_valid = false;
	return out;
	}

AudioBuffer AudioBuffer::copy() const
	{
	AudioBuffer out;
	out.format = format;
	out.buffer = get_buffer();   // deep copy of the samples
	return out;
	}

bool AudioBuffer::is_null() const
	{
	auto held = lock.hold();
	return count() == 0 || ( host_valid && buffer.empty() && !dev ) || format.sample_rate == 0;
	}

void AudioBuffer::clear_buffer()
	{
	auto held = lock.hold();
	buffer.assign( count(), 0.0f );
	host_valid = true;
	dev.reset();
	}

const std::vector<float> & AudioBuffer::get_buffer() const
	{
	auto held = lock.hold();           // concurrent const methods on one object: the first one downloads, the others wait (mirror_lock.h)
	if( !host_valid )
		{
		if( buffer.capacity() < count() )                // fresh memory: let every worker fault its share of the pages in, not this thread alone
			{
			buffer.reserve( count() );
			detail::touch_pages( buffer.data(), sizeof( float ) * count() );
			}
		buffer.resize( count() );
		if( dev && count() && !detail::download_to_host( buffer.data(), dev->ptr, sizeof( float ) * count() ) )
			std::cerr << "flan: download of audio failed: " << flanhip_last_error() << std::endl;
		host_valid = true;
		}
	return buffer;
	}

std::vector<float> & AudioBuffer::get_buffer()
	{
	std::as_const( *this ).get_buffer();
	auto held = lock.hold();
	dev.reset();                       // the caller may write: the host copy is the truth from here on
	return buffer;
	}

Sample AudioBuffer::get_sample( Channel c, Frame f ) const { return get_buffer()[get_buffer_pos( c, f )]; }
Sample & AudioBuffer::get_sample( Channel c, Frame f ) { return get_buffer()[get_buffer_pos( c, f )]; }
void AudioBuffer::set_sample( Channel c, Frame f, Sample s ) { get_buffer()[get_buffer_pos( c, f )] = s; }

bool AudioBuffer::is_nan_or_inf() const
	{
	for( float v : get_buffer() ) if( std::isnan( v ) || std::isinf( v ) ) return true;   // AudioBuffer.cpp:58-64
	return false;
	}

Sample AudioBuffer::get_max_sample_magnitude( Second start_time, Second end_time ) const
	{
	if( get_num_frames() <= 0 ) return 0;
	if( end_time == 0 ) end_time = get_length();                                  // AudioBuffer.cpp:418-420
	const Frame start_frame = std::clamp( Frame( time_to_frame( start_time ) ), 0, get_num_frames() - 1 );
	const Frame end_frame = std::clamp( Frame( time_to_frame( end_time ) ), 0, get_num_frames() - 1 );
	const std::vector<float> & data = get_buffer();
	Magnitude m = 0;
	for( Channel channel = 0; channel < get_num_channels(); ++channel )
		for( Frame frame = start_frame; frame < end_frame; ++frame )
			m = std::max( m, std::abs( data[get_buffer_pos( channel, frame )] ) );
	return m;
	}

void AudioBuffer::print_summary() const
	{
	std::cout << "\n=========================== Audio Info ==========================="       // AudioBuffer.cpp:500-509
	          << "\nChannels:\t" << get_num_channels() << "\nSamples:\t" << get_num_frames() << "\nSample Rate:\t" << get_sample_rate()
	          << "\n==================================================================" << "\n\n";
	}

std::shared_ptr<detail::DeviceBlock> AudioBuffer::device_block() const
	{
	auto held = lock.hold();
	if( !dev )
		{
		if( count() == 0 ) return nullptr;
		auto block = detail::DeviceBlock::allocate( sizeof( float ) * count() );
		if( !block ) return nullptr;
		if( !detail::upload_from_host( block->ptr, buffer.data(), sizeof( float ) * count() ) )
			{
			std::cerr << "flan: upload of audio failed: " << flanhip_last_error() << std::endl;
			return nullptr;
			}
		dev = std::move( block );
		}
	return dev;
	}

const float * AudioBuffer::device_data() const
	{
	const auto block = device_block();
	return block ? static_cast<const float*>( block->ptr ) : nullptr;
	}

} // namespace flan
