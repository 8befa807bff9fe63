// PVBuffer.cpp -- host/device mirrored PV container and the .flan file format
// (reference: src/flan/PV/PVBuffer.cpp:19-50 constructors / null / NaN scan, :99-140 save, :216-273 load,
//  src/flan/Utility/Bytes.cpp:70-118 writeRIFF).
#include "flan/PVBuffer.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <utility>

#include "device_block.h"

namespace flan {

PVBuffer::PVBuffer() : format(), buffer() {}
PVBuffer::PVBuffer( const Format & other ) : format( other ), buffer( count() ) {}
PVBuffer::PVBuffer( const std::string & filename ) : format(), buffer() { load( filename ); }

PVBuffer PVBuffer::adopt_device( const Format & f, std::shared_ptr<detail::DeviceBlock> block )
	{
	PVBuffer out;
	out.format = f;
	out.dev = std::move( block );
	out.host_valid = false;
	return out;
	}

PVBuffer PVBuffer::copy() const
	{
	PVBuffer out;
	out.format = format;
	out.buffer = get_buffer();
	return out;
	}

bool PVBuffer::is_null() const
	{
	auto held = lock.hold();
	return count() == 0 || ( host_valid && buffer.empty() && !dev ) || format.sample_rate == 0;
	}

bool PVBuffer::is_nan_or_inf() const
	{
	for( const MF & mf : get_buffer() )
		if( std::isnan( mf.m ) || std::isnan( mf.f ) || std::isinf( mf.m ) || std::isinf( mf.f ) ) return true;
	return false;
	}

void PVBuffer::clear_buffer()
	{
	auto held = lock.hold();
	buffer.assign( count(), MF{ 0.0f, 0.0f } );
	host_valid = true;
	dev.reset();
	synth_ws.reset();
	}

const std::vector<MF> & PVBuffer::get_buffer() const
	{
	// const methods may run concurrently on one object (they are pure reads in the reference): the first one brings the data over
	// under the object's lock, the others wait for it; once host_valid is set no const method touches the vector again
	auto held = lock.hold();
	if( !host_valid )
		{
		if( buffer.capacity() < count() )                // fresh memory: let every worker fault its share of the pages in, not this thread alone
			{
			buffer.reserve( count() );
			detail::touch_pages( buffer.data(), sizeof( MF ) * count() );
			}
		buffer.resize( count() );
		if( dev && count() && !detail::download_to_host( buffer.data(), dev->ptr, sizeof( MF ) * count() ) )
			std::cerr << "flan: download of PV failed: " << flanhip_last_error() << std::endl;
		host_valid = true;
		}
	return buffer;
	}

std::vector<MF> & PVBuffer::get_buffer()
	{
	std::as_const( *this ).get_buffer();
	auto held = lock.hold();
	dev.reset();
	synth_ws.reset();                  // the caller may write: anything derived from the old data is stale
	return buffer;
	}

MF PVBuffer::get_MF( Channel c, Frame f, Bin b ) const { return get_buffer()[get_buffer_pos( c, f, b )]; }
MF & PVBuffer::get_MF( Channel c, Frame f, Bin b ) { return get_buffer()[get_buffer_pos( c, f, b )]; }

std::shared_ptr<detail::DeviceBlock> PVBuffer::device_block() const
	{
	auto held = lock.hold();
	if( !dev )
		{
		if( count() == 0 ) return nullptr;
		auto block = detail::DeviceBlock::allocate( sizeof( MF ) * count() );
		if( !block ) return nullptr;
		if( !detail::upload_from_host( block->ptr, buffer.data(), sizeof( MF ) * count() ) )
			{
			std::cerr << "flan: upload of PV failed: " << flanhip_last_error() << std::endl;
			return nullptr;
			}
		dev = std::move( block );
		}
	return dev;
	}

const MF * PVBuffer::device_data() const
	{
	const auto block = device_block();
	return block ? static_cast<const MF*>( block->ptr ) : nullptr;
	}

Magnitude PVBuffer::get_max_partial_magnitude() const
	{
	float max_magnitude = 0;                                                     // PVBuffer.cpp:396-406
	for( const MF & mf : get_buffer() ) max_magnitude = std::max( max_magnitude, std::abs( mf.m ) );
	return max_magnitude;
	}

Magnitude PVBuffer::get_max_partial_magnitude( uint32_t start_frame, uint32_t end_frame, uint32_t start_bin, uint32_t end_bin ) const
	{
	if( end_frame == 0 ) end_frame = uint32_t( get_num_frames() );                // PVBuffer.cpp:410-411
	if( end_bin == 0 ) end_bin = uint32_t( get_num_bins() );
	end_frame = std::min( end_frame, uint32_t( get_num_frames() ) );              // (the reference reads past the end for larger values)
	end_bin = std::min( end_bin, uint32_t( get_num_bins() ) );
	const std::vector<MF> & data = get_buffer();
	float max_magnitude = 0;
	for( Channel channel = 0; channel < get_num_channels(); ++channel )
		for( uint32_t frame = start_frame; frame < end_frame; ++frame )
			for( uint32_t bin = start_bin; bin < end_bin; ++bin )
				max_magnitude = std::max( max_magnitude, std::abs( data[get_buffer_pos( channel, Frame( frame ), Bin( bin ) )].m ) );
	return max_magnitude;
	}

void PVBuffer::print_summary() const
	{
	std::cout << "\n=========================== PVBuffer Info ==========================="    // PVBuffer.cpp:535-548
	          << "\nChannels:\t" << get_num_channels() << "\nSamples:\t" << get_num_frames() << "\nBins:\t" << get_num_bins()
	          << "\nFrames/second:\t" << time_to_frame( 1 ) << "\nBins/Frequency:\t" << frequency_to_bin( 1 )
	          << "\nHop size:\t" << get_hop_size() << "\nDFT size:\t" << get_dft_size()
	          << "\n=======================================================================" << "\n\n";
	}

// ---- .flan files ---------------------------------------------------------------------------------------------------
namespace {
void put16( std::vector<uint8_t> & v, uint16_t x ) { v.push_back( x & 0xFF ); v.push_back( x >> 8 ); }
void put32( std::vector<uint8_t> & v, uint32_t x ) { for( int i = 0; i < 4; ++i ) v.push_back( ( x >> ( 8 * i ) ) & 0xFF ); }
void putTag( std::vector<uint8_t> & v, const char * t ) { for( int i = 0; i < 4; ++i ) v.push_back( uint8_t( t[i] ) ); }
}

bool PVBuffer::save( const std::string & filename ) const
	{
	const std::vector<MF> & data = get_buffer();
	const double limit = std::pow( 2, 8 * 3 - 1 );                               // PVBuffer.cpp:101-102
	const float window_size_f = float( get_dft_size() );                        // :103
	const float max_frequency_f = get_sample_rate();                            // :104

	std::vector<uint8_t> bytes( data.size() * 6 );
	for( size_t i = 0; i < data.size(); ++i )                                    // :108-125 (same order: channel, frame, bin)
		{
		const int32_t m_32 = int32_t( double( std::clamp( data[i].m / window_size_f, -1.0f, 1.0f ) ) * limit );
		const int32_t f_32 = int32_t( double( std::clamp( data[i].f / max_frequency_f, -1.0f, 1.0f ) ) * limit );
		uint8_t * p = bytes.data() + i * 6;
		p[0] = uint8_t( m_32 >> 0 ); p[1] = uint8_t( m_32 >> 8 ); p[2] = uint8_t( m_32 >> 16 );
		p[3] = uint8_t( f_32 >> 0 ); p[4] = uint8_t( f_32 >> 8 ); p[5] = uint8_t( f_32 >> 16 );
		}

	// Bytes.cpp:70-118 writeRIFF( filename, "PV", ... ): RIFF chunk (size field is the constant 4), fmt chunk, data chunk
	std::vector<uint8_t> head;
	putTag( head, "RIFF" ); put32( head, 4 ); putTag( head, "PV\0\0" );
	putTag( head, "fmt " ); put32( head, 2 + 2 + 4 * 6 + 2 );
	put16( head, 1 );                                        // formatting
	put16( head, uint16_t( get_num_channels() ) );
	put32( head, uint32_t( get_num_frames() ) );
	put32( head, uint32_t( get_num_bins() ) );
	put32( head, uint32_t( get_sample_rate() ) );
	put32( head, uint32_t( get_hop_size() ) );               // :134  (load() reads this field into analysis_rate, :245 -- reference asymmetry, kept)
	put32( head, uint32_t( get_window_size() ) );
	put32( head, 24 );
	put16( head, 1 );                                        // window type: hann
	putTag( head, "data" ); put32( head, uint32_t( bytes.size() ) );

	std::ofstream file( filename, std::ios::binary );
	if( !file ) { std::cout << "Error opening " + filename + " to write RIFF.\n"; return false; }
	file.write( reinterpret_cast<const char*>( head.data() ), std::streamsize( head.size() ) );
	file.write( reinterpret_cast<const char*>( bytes.data() ), std::streamsize( bytes.size() ) );
	return true;
	}

// The file's fixed-size front matter as it lies on disk (little endian, no padding): RIFF header, "fmt " chunk, "data" chunk header.
// One read, then a table of checks (the messages are the reference's, PVBuffer.cpp:216-273), then the 6-byte records in one read.
namespace {
#pragma pack( push, 1 )
struct FlanFileHead
	{
	char riff[4]; uint32_t riff_size; char kind[4];
	char fmt_tag[4]; uint32_t fmt_size;
	uint16_t formatting, channels;
	uint32_t frames, bins, sample_rate, hop_field, window_size, bit_depth;
	uint16_t window_type;
	char data_tag[4]; uint32_t data_size;
	};
#pragma pack( pop )
static_assert( sizeof( FlanFileHead ) == 58, ".flan front matter is 58 bytes" );

inline float unpack24( const uint8_t * p, float scale )
	{
	const int32_t v = int32_t( uint32_t( p[0] ) | uint32_t( p[1] ) << 8 | uint32_t( p[2] ) << 16 | ( p[2] & 0x80 ? 0xFF000000u : 0u ) );
	return float( double( v ) / 8388608.0 ) * scale;                              // / 2^23, then back to magnitude / Hz (PVBuffer.cpp:254-262)
	}
}

bool PVBuffer::load( const std::string & filename )
	{
	std::ifstream file( filename, std::ios::binary );
	if( !file ) { std::cout << "Error opening " + filename + " to load PV." << std::endl; return false; }
	FlanFileHead h{};
	file.read( reinterpret_cast<char*>( &h ), sizeof( h ) );

	const struct { bool ok; std::string message; } checks[] = {
		{ std::memcmp( h.riff, "RIFF", 4 ) == 0,     filename + " isn't a correctly formatted RIFF file.\n" },
		{ std::memcmp( h.kind, "PV\0\0", 4 ) == 0,   filename + " isn't a PV file.\n" },
		{ std::memcmp( h.fmt_tag, "fmt ", 4 ) == 0,  filename + " isn't formatted correctly (\"fmt \" wasn't at the start of the format chunk).\n" },
		{ h.formatting == 1,                         "Formatting must be 1 (signed int)." },
		{ h.bit_depth == 24,                         "Bit depth must be 24." },
		{ h.window_type == 1,                        "PV window must be 1 (hann)." },
		{ std::memcmp( h.data_tag, "data", 4 ) == 0, filename + " isn't a correctly formatted PV file (\"data\" wasn't at the start of the data chunk).\n" },
		};
	for( const auto & c : checks )
		if( !c.ok ) { std::cout << c.message << std::endl; return false; }

	Format fmt;
	fmt.num_channels = h.channels;
	fmt.num_frames = Frame( h.frames );
	fmt.num_bins = Bin( h.bins );
	fmt.sample_rate = FrameRate( h.sample_rate );
	fmt.analysis_rate = FrameRate( h.hop_field );              // the reference reads the HOP field into analysis_rate (PVBuffer.cpp:245): kept
	fmt.window_size = Frame( h.window_size );
	*this = PVBuffer( fmt );

	std::vector<uint8_t> packed( buffer.size() * 6 );
	file.read( reinterpret_cast<char*>( packed.data() ), std::streamsize( packed.size() ) );
	const size_t got = size_t( file.gcount() ) / 6;            // a short file leaves the rest of the buffer zero, as the reference's failed reads do
	const float m_scale = float( get_dft_size() ), f_scale = get_sample_rate();
	for( size_t i = 0; i < std::min( got, buffer.size() ); ++i )
		buffer[i] = MF{ unpack24( packed.data() + 6 * i, m_scale ), unpack24( packed.data() + 6 * i + 3, f_scale ) };
	return true;
	}

} // namespace flan
