// flan/AudioBuffer.h -- audio container (mirrors the reference's src/flan/Audio/AudioBuffer.h:20-39,138-228 and
// AudioBuffer.cpp:17-29,479-482 for the parts the phase-vocoder path uses; no libsndfile I/O).
//
// Layout: float[channel][frame], channel-major.  Move-only, explicit copy(), like the reference.
// MI355X addition: a buffer may live in HBM only.  Results of device algorithms stay on the device until host code asks
// for the samples (get_buffer / get_sample), so convert_to_PV -> stretch -> convert_to_audio never crosses PCIe.
#pragma once
#include <memory>
#include <vector>

#include "flan/defines.h"
#include "flan/mirror_lock.h"

namespace flan {

namespace detail { struct DeviceBlock; }

class AudioBuffer
	{
public:
	struct Format
		{
		Channel num_channels = 0;
		Frame num_frames = 0;
		FrameRate sample_rate = 48000;
		};

	AudioBuffer( const AudioBuffer & ) = delete;
	AudioBuffer( AudioBuffer && ) = default;
	AudioBuffer & operator=( const AudioBuffer & ) = delete;
	AudioBuffer & operator=( AudioBuffer && ) = default;
	~AudioBuffer() = default;

	AudioBuffer();
	explicit AudioBuffer( const Format & );                                     // zero-initialised samples (AudioBuffer.cpp:26-29)
	AudioBuffer( std::vector<float> && buffer, Channel num_channels, FrameRate ); // AudioBuffer.cpp:17-24

	AudioBuffer copy() const;
	bool is_null() const;                                                         // empty buffer or sample_rate 0
	void clear_buffer();

	Format get_format() const { return format; }
	Channel get_num_channels() const { return format.num_channels; }
	Frame get_num_frames() const { return format.num_frames; }
	FrameRate get_sample_rate() const { return format.sample_rate; }
	Second get_length() const { return format.num_frames / format.sample_rate; }
	size_t get_buffer_pos( Channel c, Frame f ) const { return size_t( c ) * format.num_frames + f; }   // AudioBuffer.cpp:479-482

	Sample get_sample( Channel c, Frame f ) const;
	Sample & get_sample( Channel c, Frame f );
	void set_sample( Channel c, Frame f, Sample s );
	// the rest of the host-side accessors (AudioBuffer.h:96,128,164-216); libsndfile I/O (load / save / play) is outside this library
	Sample * get_sample_pointer( Channel c, Frame f ) { return get_buffer().data() + get_buffer_pos( c, f ); }       // AudioBuffer.cpp:450-458
	const Sample * get_sample_pointer( Channel c, Frame f ) const { return get_buffer().data() + get_buffer_pos( c, f ); }
	std::vector<Sample>::const_iterator channel_begin( Channel c ) const { return get_buffer().begin() + std::ptrdiff_t( get_buffer_pos( c, 0 ) ); }   // :469-477
	std::vector<Sample>::const_iterator channel_end( Channel c ) const { return get_buffer().begin() + std::ptrdiff_t( get_buffer_pos( c + 1, 0 ) ); }
	Second frame_to_time( fFrame f ) const { return f / get_sample_rate(); }                                         // :401-409
	fFrame time_to_frame( Second t ) const { return t * float( get_sample_rate() ); }
	bool is_nan_or_inf() const;                                                                                      // :58-64
	Sample get_max_sample_magnitude( Second start_time = 0, Second end_time = 0 ) const;                             // :416-430
	void print_summary() const;                                                                                      // :500-509
	const std::vector<float> & get_buffer() const;                               // downloads from HBM on first use
	std::vector<float> & get_buffer();                                           // ... and drops the device copy (host now owns the truth)

	// ---- device residency (MI355X) ----
	bool is_device_resident() const { auto held = lock.hold(); return bool( dev ); }
	const float * device_data() const;                                           // uploads on first use; nullptr on failure
	static AudioBuffer adopt_device( const Format &, std::shared_ptr<detail::DeviceBlock> );
	std::shared_ptr<detail::DeviceBlock> device_block() const;                   // the shared handle on the HBM copy (uploads on first use)

protected:
	size_t count() const { return size_t( format.num_channels ) * size_t( format.num_frames ); }
	Format format;
	mutable std::vector<float> buffer;
	mutable bool host_valid = true;
	mutable std::shared_ptr<detail::DeviceBlock> dev;
	detail::MirrorLock lock;                       // guards buffer / host_valid / dev against concurrent const methods (mirror_lock.h)
	};

} // namespace flan
