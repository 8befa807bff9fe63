// flan/PVBuffer.h -- phase-vocoder container (mirrors the reference's src/flan/PV/PVBuffer.h:27-52,135-288 and
// PVBuffer.cpp:19-50,341-384,428-446,526-529).
//
// Layout: MF[channel][frame][bin].  Move-only, explicit copy().  Like AudioBuffer, the data may live in HBM only.
#pragma once
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "flan/defines.h"
#include "flan/mirror_lock.h"

namespace flan {

namespace detail { struct DeviceBlock; }

class PVBuffer
	{
public:
	struct Format
		{
		Channel num_channels = 0;
		Frame num_frames = 0;
		Bin num_bins = 0;
		FrameRate sample_rate = 48000;
		FrameRate analysis_rate = 48000 / 128;
		Frame window_size = 0;
		};

	PVBuffer( const PVBuffer & ) = delete;
	PVBuffer( PVBuffer && ) = default;
	PVBuffer & operator=( const PVBuffer & ) = delete;
	PVBuffer & operator=( PVBuffer && ) = default;
	~PVBuffer() = default;

	PVBuffer();
	explicit PVBuffer( const Format & );
	explicit PVBuffer( const std::string & filename );                           // load(), PVBuffer.cpp:24-29

	PVBuffer copy() const;
	bool is_null() const;                                                        // PVBuffer.cpp:39-42
	bool is_nan_or_inf() const;                                                  // PVBuffer.cpp:44-50
	void clear_buffer();

	// .flan RIFF files (PVBuffer.cpp:99-140 save, :216-273 load): 24-bit m / dft and f / sample_rate
	bool save( const std::string & filename ) const;
	bool load( const std::string & filename );

	Format get_format() const { return format; }
	Channel get_num_channels() const { return format.num_channels; }
	Frame get_num_frames() const { return format.num_frames; }
	Bin get_num_bins() const { return format.num_bins; }
	FrameRate get_sample_rate() const { return format.sample_rate; }
	FrameRate get_analysis_rate() const { return format.analysis_rate; }
	Frame get_window_size() const { return format.window_size; }
	Frame get_dft_size() const { return ( format.num_bins - 1 ) * 2; }                                   // PVBuffer.cpp:356-359
	Frame get_hop_size() const { return Frame( format.sample_rate / format.analysis_rate ); }           // PVBuffer.cpp:381-384
	Second get_length() const { return frame_to_time( fFrame( format.num_frames ) ); }
	Frequency get_height() const { return bin_to_frequency( fBin( format.num_bins ) ); }
	fFrame time_to_frame( Second t ) const { return t * float( get_sample_rate() ) / float( get_hop_size() ); }      // :428-431
	Second frame_to_time( fFrame f ) const { return f / ( float( get_sample_rate() ) / float( get_hop_size() ) ); }  // :433-436
	fBin frequency_to_bin( Frequency f ) const { return f / ( float( get_sample_rate() ) / float( get_dft_size() ) ); } // :438-441
	Frequency bin_to_frequency( fBin b ) const { return b * float( get_sample_rate() ) / float( get_dft_size() ); }  // :443-446
	size_t get_buffer_pos( Channel c, Frame f, Bin b ) const                                             // :526-529, 64-bit here
		{ return ( size_t( c ) * format.num_frames + f ) * format.num_bins + b; }

	MF get_MF( Channel c, Frame f, Bin b ) const;
	MF & get_MF( Channel c, Frame f, Bin b );
	const std::vector<MF> & get_buffer() const;
	std::vector<MF> & get_buffer();

	// the rest of the host-side accessors (PVBuffer.h:124,190-278): they work on the host copy (brought over on first use; the
	// non-const ones make it the truth, like get_buffer())
	void set_MF( Channel c, Frame f, Bin b, MF mf ) { get_MF( c, f, b ) = mf; }                                      // PVBuffer.cpp:472-475
	MF * get_MF_pointer( Channel c, Frame f, Bin b ) { return get_buffer().data() + get_buffer_pos( c, f, b ); }      // :486-489
	const MF * get_MF_pointer( Channel c, Frame f, Bin b ) const { return get_buffer().data() + get_buffer_pos( c, f, b ); }
	std::vector<MF>::iterator channel_begin( Channel c ) { return get_buffer().begin() + std::ptrdiff_t( get_buffer_pos( c, 0, 0 ) ); }       // :506-524
	std::vector<MF>::iterator channel_end( Channel c ) { return get_buffer().begin() + std::ptrdiff_t( get_buffer_pos( c + 1, 0, 0 ) ); }
	std::vector<MF>::const_iterator channel_begin( Channel c ) const { return get_buffer().begin() + std::ptrdiff_t( get_buffer_pos( c, 0, 0 ) ); }
	std::vector<MF>::const_iterator channel_end( Channel c ) const { return get_buffer().begin() + std::ptrdiff_t( get_buffer_pos( c + 1, 0, 0 ) ); }
	Channel bound_channel( Channel c ) const { return std::clamp( c, 0, get_num_channels() - 1 ); }                  // :453-466
	Frame bound_frame( Frame f ) const { return std::clamp( f, 0, get_num_frames() - 1 ); }
	Bin bound_bin( Bin b ) const { return std::clamp( b, 0, get_num_bins() - 1 ); }
	Frequency get_frequency_offset( Channel c, Frame f, Bin b ) const { return get_MF( c, f, b ).f - bin_to_frequency( fBin( b ) ); }   // :448-451
	Magnitude get_max_partial_magnitude() const;                                                                     // :396-406
	Magnitude get_max_partial_magnitude( uint32_t start_frame, uint32_t end_frame = 0, uint32_t start_bin = 0, uint32_t end_bin = 0 ) const;   // :408-426
	void print_summary() const;                                                                                      // :327-330, :535-548

	// ---- device residency (MI355X) ----
	bool is_device_resident() const { auto held = lock.hold(); return bool( dev ); }
	const MF * device_data() const;
	bool host_copy_is_current() const { auto held = lock.hold(); return host_valid; }                                             // false: the data lives on the device only
	static PVBuffer adopt_device( const Format &, std::shared_ptr<detail::DeviceBlock> );
	/** convert_to_PV leaves convert_to_audio's pre-pass (per-chain phase sums, in a synthesis workspace) next to the data; it is
	 *  valid while the data is untouched and is consumed by the first convert_to_audio (flanhip_*_fused in flanhip.h). */
	void attach_synthesis_workspace( std::shared_ptr<detail::DeviceBlock> ws, bool maybe = false ) const
		{ auto held = lock.hold(); synth_ws = std::move( ws ); synth_ws_maybe = maybe; }
	std::shared_ptr<detail::DeviceBlock> take_synthesis_workspace() const                                // one caller gets it, every other one nullptr
		{ auto held = lock.hold(); auto w = std::move( synth_ws ); synth_ws.reset(); return w; }
	/** true: the workspace MAY hold the pre-pass (left by modify_time / stretch, whose time map decides on the device):
	 *  convert_to_audio then goes through flanhip_synthesize_dev_fused_checked */
	bool synthesis_workspace_is_conditional() const { auto held = lock.hold(); return synth_ws_maybe; }
	/** the shared handle on the HBM copy (uploads on first use; empty on failure): keeps the block alive for as long as a caller works on it,
	 *  whatever other threads do to this object meanwhile */
	std::shared_ptr<detail::DeviceBlock> device_block() const;

protected:
	size_t count() const { return size_t( format.num_channels ) * size_t( format.num_frames ) * size_t( format.num_bins ); }
	Format format;
	mutable std::vector<MF> buffer;
	mutable bool host_valid = true;
	mutable std::shared_ptr<detail::DeviceBlock> dev;
	mutable std::shared_ptr<detail::DeviceBlock> synth_ws;
	mutable bool synth_ws_maybe = false;
	detail::MirrorLock lock;                       // guards buffer / host_valid / dev / synth_ws against concurrent const methods (mirror_lock.h)
	};

} // namespace flan
