// Audio.cpp -- construction and conversions of flan::Audio over the C ABI
// (reference: src/flan/Audio/AudioConstructors.cpp, Conversions/AudioPV.cpp:12-84, Audio/AudioConversions.cpp:14-56).
#include "flan/Audio.h"

#include <iostream>

#include "device_block.h"
#include "flan/PV.h"

namespace flan {

Audio::Audio() : AudioBuffer() {}
Audio::Audio( AudioBuffer && other ) : AudioBuffer( std::move( other ) ) {}
Audio Audio::copy() const { return AudioBuffer::copy(); }

Audio Audio::create_null()
	{
	std::cout << "Null Audio created";                        // AudioConstructors.cpp:19-23
	return Audio();
	}

Audio Audio::create_from_buffer( std::vector<float> && buffer, Channel num_channels, FrameRate sr )
	{
	return AudioBuffer( std::move( buffer ), num_channels, sr );
	}

Audio Audio::create_from_format( const AudioBuffer::Format & other ) { return AudioBuffer( other ); }

Audio Audio::create_empty_with_length( Second length, Channel num_channels, FrameRate sample_rate )
	{
	return create_empty_with_frames( Frame( length * sample_rate ), num_channels, sample_rate );
	}

Audio Audio::create_empty_with_frames( Frame num_frames, Channel num_channels, FrameRate sample_rate )
	{
	AudioBuffer::Format f;
	f.num_channels = num_channels; f.num_frames = num_frames; f.sample_rate = sample_rate;
	Audio out( ( AudioBuffer( f ) ) );
	out.clear_buffer();
	return out;
	}

PV Audio::convert_to_PV( Frame window_size, Frame hop, Frame dft_size, flan_CANCEL_ARG_CPP ) const
	{
	if( is_null() || hop < 1 || window_size < 2 ) return PV();
	if( canceller ) return PV();                               // flan_CANCEL_POINT( PV() ), AudioPV.cpp:49
	PVBuffer::Format f;                                        // AudioPV.cpp:20-27
	f.num_channels = get_num_channels();
	f.num_frames = Frame( flanhip_num_pv_frames( get_num_frames(), hop ) );
	f.num_bins = dft_size / 2 + 1;
	f.sample_rate = get_sample_rate();
	f.analysis_rate = get_sample_rate() / hop;
	f.window_size = window_size;

	const float * d_audio = device_data();
	if( !d_audio ) return PV();
	auto block = detail::DeviceBlock::allocate( sizeof( MF ) * size_t( f.num_channels ) * f.num_frames * f.num_bins );
	if( !block ) return PV();
	if( canceller ) return PV();
	// fused round trip: the analysis kernel also leaves what convert_to_audio's pre-pass would compute (flanhip.h)
	std::shared_ptr<detail::DeviceBlock> ws;
	const size_t ws_bytes = flanhip_synthesize_workspace_bytes( f.num_channels, f.num_frames, f.num_bins, f.sample_rate, f.analysis_rate, f.window_size );
	if( ws_bytes ) ws = detail::DeviceBlock::allocate( ws_bytes );
	const int rc = ws
		? flanhip_analyze_dev_fused( d_audio, f.num_channels, get_num_frames(), get_sample_rate(), window_size, hop, dft_size,
			static_cast<flanhip_MF*>( block->ptr ), ws->ptr, nullptr )
		: flanhip_analyze_dev( d_audio, f.num_channels, get_num_frames(), get_sample_rate(), window_size, hop, dft_size,
			static_cast<flanhip_MF*>( block->ptr ), nullptr );
	if( !detail::report( rc, "convert_to_PV" ) ) return PV();
	// flan_CANCEL_POINT while the kernels run: the flag is polled during the wait and stops the launch (flanhip_wait_cancellable_fn)
	const int waited = flanhip_wait_cancellable_fn( nullptr, detail::poll_canceller, &canceller );
	if( waited == FLANHIP_ERR_CANCELLED || canceller ) return PV();
	if( !detail::report( waited, "convert_to_PV" ) ) return PV();
	PVBuffer out = PVBuffer::adopt_device( f, std::move( block ) );
	if( ws ) out.attach_synthesis_workspace( std::move( ws ) );
	return out;
	}

PV Audio::convertToPV( Frame window_size, Frame hop, Frame dft_size, flan_CANCEL_ARG_CPP ) const
	{
	return convert_to_PV( window_size, hop, dft_size, canceller );
	}

PV Audio::convert_to_ms_PV( Frame window_size, Frame hop, Frame dft_size, flan_CANCEL_ARG_CPP ) const
	{
	if( get_num_channels() != 2 ) return PV();                 // AudioPV.cpp:82
	return convert_to_mid_side().convert_to_PV( window_size, hop, dft_size, canceller );
	}

Audio Audio::convert_to_mid_side() const
	{
	if( is_null() ) return Audio::create_null();
	if( get_num_channels() != 2 )
		{
		std::cout << "Can't transform non-stereo Audio between Mid-Side and Left-Right formats." << std::endl;   // AudioConversions.cpp:38
		return copy();
		}
	const float * d_in = device_data();
	if( !d_in ) return Audio::create_null();
	auto block = detail::DeviceBlock::allocate( sizeof( float ) * 2 * size_t( get_num_frames() ) );
	if( !block ) return Audio::create_null();
	if( !detail::report( flanhip_mid_side_dev( d_in, get_num_frames(), static_cast<float*>( block->ptr ), nullptr ), "convert_to_mid_side" ) )
		return Audio::create_null();
	flanhip_stream_synchronize( nullptr );
	return AudioBuffer::adopt_device( get_format(), std::move( block ) );
	}

Audio Audio::convert_to_left_right() const { return convert_to_mid_side(); }   // AudioConversions.cpp:53-56

Audio Audio::resample( FrameRate new_sample_rate ) const
	{
	if( is_null() ) return Audio::create_null();
	if( new_sample_rate == get_sample_rate() ) return copy();  // AudioConversions.cpp:18-19
	AudioBuffer::Format f = get_format();                      // :21-23
	f.num_frames = Frame( flanhip_resample_out_frames( get_num_frames(), get_sample_rate(), new_sample_rate ) );
	f.sample_rate = new_sample_rate;
	if( f.num_frames <= 0 ) return Audio::create_null();
	const float * d_in = device_data();
	auto block = detail::DeviceBlock::allocate( sizeof( float ) * size_t( f.num_channels ) * f.num_frames );
	if( !d_in || !block ) return Audio::create_null();
	// r8brain CDSPResampler with default parameters, one stream over the whole buffer (:25-27), whatever chain of stages it builds for the two rates
	if( !detail::report( flanhip_resample_dev( d_in, get_num_channels(), get_num_frames(), get_sample_rate(), new_sample_rate,
			static_cast<float*>( block->ptr ), nullptr ), "resample" ) ) return Audio::create_null();
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "resample" ) ) return Audio::create_null();
	return AudioBuffer::adopt_device( f, std::move( block ) );
	}

} // namespace flan
