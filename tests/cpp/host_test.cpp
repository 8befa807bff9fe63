// host_test.cpp -- exercises the C++ drop-in surface (include/flan/*.h) the way a Flan user program does
// (cf. the reference's tests/flanTest.cpp:32-47: load -> convert_to_PV -> repitch(lambda) -> convert_to_audio).
//   host_test              full run, needs an MI355X
//   host_test --no-device  what must hold without a GPU: null objects, no crash, no CPU fallback
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <iostream>
#include <string>
#include <thread>
#include <chrono>
#include <type_traits>
#include <utility>
#include <vector>

#include "flan/flan.h"
#include "flanhip.h"

using namespace flan;

static int failures = 0;
#define CHECK( cond ) do { if( !( cond ) ) { std::printf( "FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond ); ++failures; } } while( 0 )
static bool close_to( double a, double b, double rel ) { return std::fabs( a - b ) <= rel * std::fabs( b ) + 1e-30; }

static_assert( !std::is_copy_constructible_v<Audio> && std::is_move_constructible_v<Audio>, "Audio is move-only (AudioBuffer.h:23-27)" );
static_assert( !std::is_copy_constructible_v<PV> && std::is_move_constructible_v<PV>, "PV is move-only (PVBuffer.h:30-34)" );
static_assert( !std::is_copy_constructible_v<Function<TF, float>>, "Function is move-only (Function.h:42-46)" );

static Audio sine( int n )
	{
	std::vector<float> x( n );
	for( int i = 0; i < n; ++i ) x[i] = float( 0.5 * std::sin( 2.0 * 3.14159265358979323846 * 440.0 * i / 48000.0 ) );
	return Audio::create_from_buffer( std::move( x ), 1, 48000.0f );
	}

static uint32_t hash32( uint32_t x ) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static Audio noise( int ch, int n, uint32_t seed )
	{
	std::vector<float> x( size_t( ch ) * n );
	for( int c = 0; c < ch; ++c ) for( int i = 0; i < n; ++i )
		x[size_t( c ) * n + i] = float( hash32( hash32( seed ^ ( uint32_t( c ) * 0x9E3779B9U ) ) + uint32_t( i ) * 0x85EBCA6BU ) >> 8 ) * ( 1.0f / 8388608.0f ) - 1.0f;
	return Audio::create_from_buffer( std::move( x ), ch, 48000.0f );
	}

// the worker pool and the staging memory behind Function::sample (flan_amd/host/host_runtime.cpp); runs with or without a device
static void host_runtime_checks()
	{
	CHECK( detail::host_workers() >= 1 );
	// a large grid sampled in parallel = the same grid sampled sequentially, element for element, and no element left untouched
	Function<TF, float> par( []( TF tf ){ return tf.t * 3.0f + tf.f; } );
	Function<TF, float> seq( []( TF tf ){ return tf.t * 3.0f + tf.f; }, ExecutionPolicy::Linear_Sequenced );
	for( int frames : { 1, 15, 16, 17, 129, 1000 } )
		{
		const auto a = par.sample( 0, float( frames ), 0.01f, 0, 513, 23.4375f );
		const auto b = seq.sample( 0, float( frames ), 0.01f, 0, 513, 23.4375f );
		CHECK( a.size() == size_t( frames ) * 513 && a.get_vector().size() == a.size() );
		CHECK( std::memcmp( a.get_vector().data(), b.get_vector().data(), sizeof( float ) * a.size() ) == 0 );
		}
	// every index exactly once, whatever the range
	for( int n : { 0, 1, 127, 128, 129, 5000, 100003 } )
		{
		std::vector<std::atomic<int>> hits( size_t( n ) + 1 );
		detail::for_each_index( 7, 7 + n, ExecutionPolicy::Parallel_Unsequenced, [&]( int i ){ hits[size_t( i - 7 )].fetch_add( 1 ); } );
		bool once = true;
		for( int i = 0; i < n; ++i ) once = once && hits[size_t( i )].load() == 1;
		CHECK( once && hits[size_t( n )].load() == 0 );
		}
	// a parallel region started from inside one runs inline instead of deadlocking
	std::atomic<long> total{ 0 };
	detail::for_each_index( 0, 256, ExecutionPolicy::Parallel_Unsequenced, [&]( int )
		{ detail::for_each_index( 0, 256, ExecutionPolicy::Parallel_Unsequenced, [&]( int ){ total.fetch_add( 1 ); } ); } );
	CHECK( total.load() == 256 * 256 );
	// what a callable throws reaches the caller, and the pool is usable afterwards
	bool caught = false;
	try { detail::for_each_index( 0, 4096, ExecutionPolicy::Parallel_Unsequenced, []( int i ){ if( i == 1234 ) throw std::runtime_error( "boom" ); } ); }
	catch( const std::runtime_error & ) { caught = true; }
	CHECK( caught );
	total = 0;
	detail::for_each_index( 0, 4096, ExecutionPolicy::Parallel_Unsequenced, [&]( int ){ total.fetch_add( 1 ); } );
	CHECK( total.load() == 4096 );
	// staging blocks: reusable, writable end to end, large and small
	for( size_t bytes : { size_t( 1 ), size_t( 4096 ), size_t( 300 ) << 10, size_t( 5 ) << 20 } )
		for( int rep = 0; rep < 3; ++rep )
			{
			auto * p = static_cast<unsigned char*>( detail::staging_acquire( bytes ) );
			CHECK( p != nullptr );
			std::memset( p, 0xA5, bytes );
			CHECK( p[0] == 0xA5 && p[bytes - 1] == 0xA5 );
			detail::staging_release( p );
			}
	}

static void no_device_checks()
	{
	CHECK( flanhip_device_count() == 0 );
	Audio a = sine( 4800 );
	CHECK( !a.is_null() && a.get_num_frames() == 4800 && a.get_num_channels() == 1 );
	PV pv = a.convert_to_PV( 2048, 512, 2048 );       // no device: the path fails loudly and returns a null object
	CHECK( pv.is_null() );
	CHECK( std::strstr( flanhip_last_error(), "no HIP device" ) != nullptr );
	CHECK( pv.convert_to_audio().is_null() );
	CHECK( pv.stretch( 2.0f ).is_null() );
	CHECK( Audio().convert_to_PV().is_null() );
	// host-side pieces that need no device
	CHECK( flanhip_num_pv_frames( 240000, 512 ) == 469 );
	CHECK( flanhip_hop_size( 48000.0f, 93.75f ) == 512 );
	PVBuffer::Format f; f.num_channels = 1; f.num_frames = 3; f.num_bins = 5; f.sample_rate = 48000; f.analysis_rate = 48000.0f / 512; f.window_size = 8;
	PV p = PV::create_from_format( f );
	CHECK( !p.is_null() && p.get_dft_size() == 8 && p.get_hop_size() == 512 );
	CHECK( p.bin_to_frequency( 1 ) == 6000.0f && p.frequency_to_bin( 12000.0f ) == 2.0f );
	CHECK( close_to( p.time_to_frame( 1.0f ), 93.75, 1e-7 ) );
	p.get_MF( 0, 1, 2 ) = MF{ 3.0f, 440.0f };
	const std::string path = "/tmp/flan_host_test.flan";
	CHECK( p.save( path ) );
	PV q = PV::load_from_file( path );
	CHECK( q.get_num_frames() == 3 && q.get_num_bins() == 5 && q.get_window_size() == 8 );
	CHECK( q.get_analysis_rate() == 512.0f );         // the reference's save/load asymmetry: the hop lands in analysis_rate (PVBuffer.cpp:134,245)
	CHECK( close_to( q.get_MF( 0, 1, 2 ).m, 3.0, 1e-6 ) && close_to( q.get_MF( 0, 1, 2 ).f, 440.0, 1e-4 ) );
	// the small accessors of the buffer classes (PVBuffer.h:190-278, AudioBuffer.h:96-216)
	p.set_MF( 0, 2, 4, MF{ -7.0f, 23999.0f } );
	CHECK( p.get_MF_pointer( 0, 2, 4 )->m == -7.0f && p.get_max_partial_magnitude() == 7.0f && p.get_max_partial_magnitude( 0, 2 ) == 3.0f );
	CHECK( p.channel_end( 0 ) - p.channel_begin( 0 ) == 15 && p.bound_bin( 99 ) == 4 && p.bound_frame( -3 ) == 0 && p.bound_channel( 5 ) == 0 );
	CHECK( p.get_frequency_offset( 0, 1, 2 ) == 440.0f - 12000.0f );
	p.set_MF( 0, 1, 3, MF{ 5.0f, 500.0f } );
	CHECK( p.getBinInterpolated( 0, Frame( 1 ), 2.5f ).m == 0.5f * 3.0f + 0.5f * 5.0f && p.getBinInterpolated( 0, 1.0f, Bin( 2 ) ).f == 440.0f );
	CHECK( p.getBinInterpolated( 0, 0.5f, 2.5f ).m == 0.5f * ( 0.5f * 0.0f + 0.5f * 3.0f ) + 0.5f * ( 0.5f * 0.0f + 0.5f * 5.0f ) );
	CHECK( p.sample_function_over_time_domain( Function<Second, float>( []( Second t ){ return t * 2.0f; } ) ).size() == 3 );
	Audio tiny = Audio::create_from_buffer( { 0.5f, -2.0f, 0.25f, 1.0f, 0.0f, -0.75f }, 2, 48000.0f );
	CHECK( tiny.get_max_sample_magnitude() == 2.0f && !tiny.is_nan_or_inf() && tiny.time_to_frame( 0.5f ) == 24000.0f && tiny.frame_to_time( 48.0f ) == 0.001f );
	CHECK( *tiny.get_sample_pointer( 1, 0 ) == 1.0f && tiny.channel_end( 1 ) - tiny.channel_begin( 1 ) == 3 );
	tiny.set_sample( 0, 0, std::nanf( "" ) );
	CHECK( tiny.is_nan_or_inf() );
	Function<TF, float> c( 2.0f );
	CHECK( c.is_constant() && c( TF{ 0, 0 } ) == 2.0f );
	Function<TF, float> l( []( TF tf ){ return tf.t + tf.f; } );
	auto s = p.sample_function_over_domain( l );
	CHECK( !s.is_constant() && s.size() == 15 && close_to( s.at( 2, 3 ), 2.0 / 93.75 + 18000.0, 1e-6 ) );
	host_runtime_checks();
	}

static void device_checks()
	{
	host_runtime_checks();                              // with a device the large staging blocks are page-locked
	CHECK( flanhip_device_count() >= 1 );
	// ---- BASELINE config 1 + SURVEY 8c anchors through the class surface
	Audio a = sine( 240000 );
	PV pv = a.convert_to_PV( 2048, 512, 2048 );
	CHECK( !pv.is_null() && pv.is_device_resident() );
	CHECK( pv.get_num_channels() == 1 && pv.get_num_frames() == 469 && pv.get_num_bins() == 1025 );
	CHECK( pv.get_window_size() == 2048 && pv.get_hop_size() == 512 && pv.get_dft_size() == 2048 && pv.get_analysis_rate() == 93.75f );
	CHECK( close_to( pv.get_MF( 0, 100, 19 ).m, 247.517, 2e-5 ) && close_to( pv.get_MF( 0, 100, 19 ).f, 440.0, 2e-6 ) );
	CHECK( close_to( pv.get_MF( 0, 0, 19 ).m, 126.01691, 2e-5 ) && close_to( pv.get_MF( 0, 0, 19 ).f, 489.28607, 2e-6 ) );
	Audio back = pv.convert_to_audio();
	CHECK( back.get_num_frames() == 240128 && back.get_sample_rate() == 48000.0f );
	CHECK( close_to( back.get_sample( 0, 1000 ), 0.43333316, 1e-5 ) );
	double e = 0; for( float v : back.get_buffer() ) e += double( v ) * v;
	CHECK( close_to( e, 30006.053, 1e-5 ) );
	CHECK( !a.convertToPV( 2048, 512, 2048 ).convertToAudio().is_null() );   // older spellings
	// the first convert_to_audio of a fresh PV takes the fused pre-pass, the second recomputes it: same audio either way
	Audio back2 = pv.convert_to_audio();
	CHECK( std::memcmp( back.get_buffer().data(), back2.get_buffer().data(), sizeof( float ) * back.get_buffer().size() ) == 0 );
	// default arguments are the reference's: window 2048, hop 128, dft 4096 (Audio.h:158-163)
	PV dflt = sine( 20000 ).convert_to_PV();
	CHECK( dflt.get_num_bins() == 2049 && dflt.get_hop_size() == 128 && dflt.get_num_frames() == 20000 / 128 + 1 );
	Audio dflt_back = dflt.convert_to_audio();                                     // dft 4096, hop 128: the API's default shape
	CHECK( dflt_back.get_num_frames() == dflt.get_num_frames() * 128 );
	CHECK( close_to( dflt_back.get_sample( 0, 4000 ), sine( 20000 ).get_sample( 0, 4000 ) * 1.00074, 2e-3 ) );

	// ---- frame processors, chained on the device
	Audio n2 = noise( 2, 60000, 1234 );
	PV p2 = n2.convert_to_PV( 2048, 512, 2048 );
	PV st_l = p2.stretch( []( TF ){ return 2.0f; } );
	PV st_c = p2.stretch( 2.0f );                     // a constant behaves like the callable (not like the reference's doubling bug)
	CHECK( st_l.get_num_frames() == 2 * p2.get_num_frames() && st_c.get_num_frames() == st_l.get_num_frames() );
	CHECK( std::memcmp( st_l.get_buffer().data(), st_c.get_buffer().data(), sizeof( MF ) * st_l.get_buffer().size() ) == 0 );
	Audio st_audio = st_l.convert_to_audio();
	CHECK( st_audio.get_num_frames() == st_l.get_num_frames() * 512 );
	PV rp = p2.repitch( []( TF ){ return 2.0f; } );
	CHECK( rp.get_num_frames() == p2.get_num_frames() && !rp.is_null() );
	PV mt = p2.modify_time( []( TF tf ){ return tf.t * 0.5f; } );
	CHECK( !mt.is_null() && mt.get_num_frames() == Frame( std::ceil( p2.time_to_frame( 0.5f * ( p2.get_num_frames() - 1 ) / 93.75f ) ) ) );
	PV mfq = p2.modify_frequency( []( TF tf ){ return tf.f * 1.5f; } );
	CHECK( !mfq.is_null() );
	PV sh_l = p2.shape( []( MF mf ){ return MF{ mf.m, mf.f + 100.0f }; } );
	PV sh_a = p2.shape_affine( 1.0f, 0.0f, 1.0f, 100.0f );
	CHECK( std::memcmp( sh_l.get_buffer().data(), sh_a.get_buffer().data(), sizeof( MF ) * sh_l.get_buffer().size() ) == 0 );
	PV sh_al = p2.shape( []( MF mf ){ return MF{ mf.m, mf.f * 2.0f }; }, true );
	PV sh_aa = p2.shape_affine( 1.0f, 0.0f, 2.0f, 0.0f, true );
	CHECK( std::memcmp( sh_al.get_buffer().data(), sh_aa.get_buffer().data(), sizeof( MF ) * sh_al.get_buffer().size() ) == 0 );
		{
		// an Interpolator built from a callable runs from a sampled table: the identity callable must give what linear() gives
		PV st_named = p2.stretch( 2.0f );
		PV st_call = p2.stretch( 2.0f, Interpolator( []( float x ){ return x; } ) );
		CHECK( !st_call.is_null() && st_call.get_num_frames() == st_named.get_num_frames() );
		size_t same = 0;
		for( size_t i = 0; i < st_named.get_buffer().size(); ++i ) same += std::memcmp( &st_named.get_buffer()[i], &st_call.get_buffer()[i], sizeof( MF ) ) == 0;
		CHECK( same >= st_named.get_buffer().size() * 999 / 1000 );
		CHECK( !p2.stretch( 2.0f, Interpolator( []( float x ){ return x * x; } ) ).is_null() );
		CHECK( !p2.repitch( 1.5f, Interpolator( []( float x ){ return x * x; } ) ).is_null() );
		}

	// ---- further frame processors (PV.cpp:205-264, :552-641; PVModify.cpp:445-511, :607-666)
		{
		const Frame F = p2.get_num_frames();
		const Bin B = p2.get_num_bins();
		PV other = noise( 2, 40000, 99 ).convert_to_PV( 2048, 512, 2048 );       // fewer frames than p2
		PV rep = p2.replace_amplitudes( other );                                  // amount defaults to 1 (PV.h:406)
		CHECK( rep.get_num_frames() == F && rep.get_MF( 1, 5, 9 ).m == other.get_MF( 1, 5, 9 ).m * 1.0f + p2.get_MF( 1, 5, 9 ).m * 0.0f );
		CHECK( rep.get_MF( 1, 5, 9 ).f == p2.get_MF( 1, 5, 9 ).f && rep.get_MF( 0, F - 1, 3 ).m == 0.0f );   // outside the overlap: cleared
		PV rep_l = p2.replace_amplitudes( other, []( TF tf ){ return tf.t; } );
		CHECK( close_to( rep_l.get_MF( 0, 40, 7 ).m, other.get_MF( 0, 40, 7 ).m * ( 40 / 93.75f ) + p2.get_MF( 0, 40, 7 ).m * ( 1.0f - 40 / 93.75f ), 1e-6 ) );
		PV sub = p2.subtract_amplitudes( p2 );
		CHECK( sub.get_MF( 0, 10, 10 ).m == 0.0f && sub.get_MF( 0, 10, 10 ).f == p2.get_MF( 0, 10, 10 ).f );
		PV keep = p2.retain_n_loudest_partials( 10 ), drop = p2.remove_n_loudest_partials( 10 );
		int kept = 0, dropped = 0;
		for( Bin b = 0; b < B; ++b ) { kept += keep.get_MF( 0, 20, b ).m != 0.0f; dropped += drop.get_MF( 0, 20, b ).m == 0.0f; }
		CHECK( kept == 10 && dropped == 10 );
		PV keep_l = p2.retain_n_loudest_partials( []( Second t ){ return Bin( t * 100.0f ); } );
		kept = 0;
		for( Bin b = 0; b < B; ++b ) kept += keep_l.get_MF( 1, 50, b ).m != 0.0f;
		CHECK( kept == Bin( p2.frame_to_time( 1 ) * 50 * 100.0f ) );
		PV res = p2.resonate( 0.5f, 0.5f );
		CHECK( res.get_num_frames() == F + Frame( std::ceil( 0.5f * 93.75f ) ) );
		const float d = std::pow( 0.5f, 1.0f / 93.75f );
		CHECK( res.get_MF( 0, F + 5, 12 ).m == res.get_MF( 0, F + 4, 12 ).m * d && res.get_MF( 0, F + 5, 12 ).f == res.get_MF( 0, F + 4, 12 ).f );
		CHECK( !p2.resonate( 0.1f, []( TF tf ){ return tf.f > 1000.0f ? 0.9f : 0.1f; } ).is_null() );
		PV des = p2.desample( 0.25f );
		CHECK( des.get_MF( 0, 3, 30 ).m == p2.get_MF( 0, 3, 30 ).m * 1.0f + 0.0f * p2.get_MF( 0, 7, 30 ).m );          // frames 0, 3, 7, ... are selected
		CHECK( des.get_MF( 0, 5, 30 ).m == 0.5f * p2.get_MF( 0, 3, 30 ).m + 0.5f * p2.get_MF( 0, 7, 30 ).m );
		CHECK( p2.desample( 0.25f, Interpolator::ceil() ).get_MF( 0, 4, 30 ).m == 0.0f * p2.get_MF( 0, 3, 30 ).m + 1.0f * p2.get_MF( 0, 7, 30 ).m );
		// frames 3 and 7 are selected, frame 5 sits half way: the callable x^2 gives mix 0.25 (exactly a sample point of the table)
		CHECK( p2.desample( 0.25f, Interpolator( []( float x ){ return x * x; } ) ).get_MF( 0, 5, 30 ).m == 0.75f * p2.get_MF( 0, 3, 30 ).m + 0.25f * p2.get_MF( 0, 7, 30 ).m );
		PV ext = p2.time_extrapolate( 0.2f, 0.6f, 0.5f );
		const Frame sf = Frame( p2.time_to_frame( 0.2f ) ), ef = Frame( p2.time_to_frame( 0.6f ) );
		CHECK( ext.get_num_frames() == ef + Frame( p2.time_to_frame( 0.5f ) ) );
		CHECK( std::memcmp( ext.get_buffer().data(), p2.get_buffer().data(), sizeof( MF ) * size_t( sf ) * B ) == 0 );   // frames before start: copied
		CHECK( !p2.time_extrapolate( 0.2f, -1, 0.5f, Interpolator( []( float x ){ return x * x; } ) ).is_null() );        // default end, any interpolator
		CHECK( p2.time_extrapolate( 0.6f, 0.2f, 0.5f ).is_null() && p2.time_extrapolate( 0.2f, 0.6f, 0.0f ).is_null() );   // PVModify.cpp:616-617
		}

	// ---- mid/side
	PV ms = n2.convert_to_ms_PV( 2048, 512, 2048 );
	CHECK( !ms.is_null() );
	Audio lr = ms.convert_to_lr_audio();
	CHECK( lr.get_num_channels() == 2 && !lr.is_null() );
	CHECK( sine( 1000 ).convert_to_ms_PV().is_null() );                            // mono: null (AudioPV.cpp:82)
	Audio m = n2.convert_to_mid_side();
	CHECK( close_to( m.get_sample( 0, 10 ), ( n2.get_sample( 0, 10 ) + n2.get_sample( 1, 10 ) ) / std::sqrt( 2.0f ), 1e-6 ) );

	// ---- resample (r8brain path): its single-step ratios on the device, other ratios refused with a null Audio
		{
		std::vector<float> x( 2 * 19200 );
		for( size_t i = 0; i < x.size(); ++i ) x[i] = float( hash32( uint32_t( i ) ) >> 8 ) * ( 1.0f / 8388608.0f ) - 1.0f;
		Audio a96 = Audio::create_from_buffer( std::move( x ), 2, 96000.0f );
		Audio a48 = a96.resample( 48000.0f );
		CHECK( !a48.is_null() && a48.get_num_frames() == 9600 && a48.get_sample_rate() == 48000.0f && a48.get_num_channels() == 2 );
		CHECK( a96.resample( 96000.0f ).get_num_frames() == 19200 );               // same rate: a copy (AudioConversions.cpp:18-19)
		CHECK( a96.resample( 1000.0f ).get_num_frames() == 200 );                 // 96x down: five third-band half-band stages and a 3:1 convolver
		CHECK( a96.resample( 11026.0f ).get_num_frames() == 2205 );               // half-band stages in front of the spline bank
		CHECK( a96.resample( 44101.0f ).get_num_frames() == 8820 );               // no whole stepping: the spline-interpolated bank
		CHECK( a96.resample( 16000.0f ).get_num_frames() == 3200 );               // half-band downsampler + block convolver
		Audio a441 = a96.resample( 44100.0f );                                    // block convolver + whole-stepping interpolator
		CHECK( !a441.is_null() && a441.get_num_frames() == 8820 && a441.get_sample_rate() == 44100.0f );
		Audio a192 = a96.resample( 192000.0f );                                   // 1:2, 3:2 ... are single block-convolver steps too
		CHECK( !a192.is_null() && a192.get_num_frames() == 38400 && a192.get_sample_rate() == 192000.0f );
		CHECK( a96.resample( 64000.0f ).get_num_frames() == 12800 );
		}

	// ---- a long chain stays in HBM end to end: nothing below touches get_buffer() until the final check
		{
		std::vector<float> x( 2 * 96000 );
		for( size_t i = 0; i < x.size(); ++i ) x[i] = float( hash32( uint32_t( i ) * 2654435761u ) >> 8 ) * ( 1.0f / 8388608.0f ) - 1.0f;
		Audio in = Audio::create_from_buffer( std::move( x ), 2, 96000.0f );
		PV chain = in.resample( 48000.0f ).convert_to_ms_PV( 2048, 512, 2048 );
		CHECK( chain.is_device_resident() );
		PV shaped = chain.stretch( 1.5f ).repitch( []( TF tf ){ return tf.t < 0.5f ? 1.0f : 1.25f; } ).retain_n_loudest_partials( 200 )
			.resonate( 0.25f, 0.6f ).shape( []( MF mf ){ return MF{ mf.m * 0.5f, mf.f }; } );
		CHECK( !shaped.is_null() && shaped.get_num_channels() == 2 );
		Audio result = shaped.convert_to_lr_audio();
		CHECK( !result.is_null() && result.get_sample_rate() == 48000.0f && result.get_num_channels() == 2 );
		CHECK( result.get_num_frames() == shaped.get_num_frames() * 512 );
		double energy = 0; bool finite = true;
		for( float v : result.get_buffer() ) { energy += double( v ) * v; finite = finite && std::isfinite( v ); }
		CHECK( finite && energy > 1.0 );
		}

	// ---- error behaviour: null objects, never exceptions
	CHECK( Audio().convert_to_PV().is_null() );
	CHECK( a.convert_to_PV( 2048, 512, 3001 ).is_null() );                          // an odd dft size
	CHECK( !a.convert_to_PV( 2048, 512, 3000 ).is_null() );                         // any even one is served (FFTHelper.cpp:16-26 hands it to FFTW)
	CHECK( a.convert_to_PV( 4096, 512, 2048 ).is_null() );                          // window larger than dft
	CHECK( PV().convert_to_audio().is_null() );
	std::atomic<bool> cancel( true );
	CHECK( a.convert_to_PV( 2048, 512, 2048, cancel ).is_null() );                  // flan_CANCEL_POINT
	CHECK( pv.convert_to_audio( cancel ).is_null() );
	// NaN in the PV: warning on stdout, result still produced (AudioPV.cpp:88-89)
	PV bad = p2.copy();
	bad.get_MF( 0, 3, 7 ).f = std::nanf( "" );
	CHECK( bad.is_nan_or_inf() && !bad.convert_to_audio().is_null() );
	// host writes invalidate the device mirror
	PV w = p2.copy();
	w.get_MF( 0, 0, 0 ) = MF{ 1.0f, 2.0f };
	CHECK( !w.is_device_resident() && w.get_MF( 0, 0, 0 ).m == 1.0f );
	// ---- selecting, rearranging and re-placing frames (PV.cpp:24-39, :92-198, :362-419, :643-727)
		{
		const Frame F = p2.get_num_frames(); const Bin B = p2.get_num_bins();
		PV fr = p2.get_frame( p2.frame_to_time( 20 ) );
		CHECK( fr.get_num_frames() == 1 && fr.get_num_channels() == 2 && fr.get_MF( 1, 0, 33 ).m == p2.get_MF( 1, 20, 33 ).m && fr.get_MF( 1, 0, 33 ).f == p2.get_MF( 1, 20, 33 ).f );
		PV half = p2.get_frame( p2.frame_to_time( 20.5f ) );
		CHECK( half.get_MF( 0, 0, 7 ).m == 0.5f * p2.get_MF( 0, 20, 7 ).m + 0.5f * p2.get_MF( 0, 21, 7 ).m );
		CHECK( p2.get_frame( 1e9f ).get_MF( 0, 0, 5 ).m == p2.get_MF( 0, F - 1, 5 ).m && p2.get_frame( -3.0f ).get_MF( 0, 0, 5 ).m == p2.get_MF( 0, 0, 5 ).m );
		PV cut = p2.cut_frames( 10, 30 );
		CHECK( cut.get_num_frames() == 20 && cut.get_MF( 1, 0, 9 ).f == p2.get_MF( 1, 10, 9 ).f && cut.get_MF( 0, 19, 100 ).m == p2.get_MF( 0, 29, 100 ).m );
		CHECK( p2.cut_frames( 30, 10 ).is_null() && p2.cut_frames( 5, 5 ).is_null() );
		CHECK( p2.cut_frames( 0, F + 100 ).get_num_frames() == F - 1 );               // the end clamps to F-1 (PV.cpp:653)
		auto pieces = p2.split_at_times( { p2.frame_to_time( 40 ), p2.frame_to_time( 15 ), -1.0f, 1e9f } );
		CHECK( pieces.size() == 3 && pieces[0].get_num_frames() == 15 && pieces[1].get_num_frames() == 25 && pieces[2].get_num_frames() == F - 1 - 40 );
		PV joined = PV::join( pieces );
		CHECK( joined.get_num_frames() == F - 1 && joined.get_num_channels() == 2 );
		CHECK( std::memcmp( joined.get_buffer().data(), p2.cut_frames( 0, F ).get_buffer().data(), sizeof( MF ) * joined.get_buffer().size() ) == 0 );
		CHECK( PV::join( std::vector<const PV *>() ).is_null() );
		PV fz = p2.freeze( { p2.frame_to_time( 12 ) }, { p2.frame_to_time( 6 ) } );
		CHECK( fz.get_num_frames() == F + 6 && fz.get_MF( 0, 12, 50 ).m == p2.get_MF( 0, 12, 50 ).m && fz.get_MF( 0, 17, 50 ).m == p2.get_MF( 0, 12, 50 ).m );
		CHECK( fz.get_MF( 0, 18, 50 ).m == p2.get_MF( 0, 13, 50 ).m && fz.get_MF( 1, F + 4, 50 ).m == p2.get_MF( 1, F - 1, 50 ).m && fz.get_MF( 1, F + 5, 50 ).m == 0.0f );
		CHECK( p2.freeze( { 0.1f, 0.2f }, { 0.1f } ).is_null() );                     // sizes differ: the reference's error
		// a selector that reads everything from one second earlier, one octave down
		PV sel = p2.select( p2.get_length(), []( TF tf ){ return TF{ tf.t - 0.1f, tf.f * 0.5f }; } );
		CHECK( sel.get_num_frames() == Frame( p2.time_to_frame( p2.get_length() ) ) && !sel.is_null() );
			{
			const Frame fo = 60; const Bin bo = 40;
			const TF s{ fo * ( 1.0f / p2.get_analysis_rate() ) - 0.1f, bo * p2.bin_to_frequency( 1 ) * 0.5f };
			const MF src = p2.get_MF( 1, Frame( p2.time_to_frame( s.t ) ), Bin( p2.frequency_to_bin( s.f ) ) );
			CHECK( sel.get_MF( 1, fo, bo ).m == src.m && sel.get_MF( 1, fo, bo ).f == src.f * ( p2.bin_to_frequency( float( bo ) ) / s.f ) );
			CHECK( sel.get_MF( 0, 2, bo ).m == 0.0f );                                 // reads from before the start: left empty
			}
		CHECK( p2.select( 0.0f, []( TF tf ){ return tf; } ).is_null() );
		// overtones of the 440 Hz sine: the octave lands in the bin of 880 Hz, from the loudest of the bins that claim it
		PV oct = pv.add_octaves( []( std::pair<Second, Harmonic> ){ return 1.0f; } );
		const float f19 = pv.get_MF( 0, 100, 19 ).f;
		CHECK( oct.get_num_frames() == pv.get_num_frames() && oct.get_MF( 0, 100, Bin( pv.frequency_to_bin( f19 * 2.0f ) ) ).f == f19 * 2.0f );
		CHECK( oct.get_MF( 0, 100, Bin( pv.frequency_to_bin( f19 * 2.0f ) ) ).m == pv.get_MF( 0, 100, 19 ).m );
		PV har = pv.add_harmonics( []( std::pair<Second, Harmonic> th ){ return th.second == 1 ? 0.25f : 0.0f; } );   // harmonic index 1 = the third partial (PV.cpp:393,417)
		CHECK( har.get_MF( 0, 100, Bin( pv.frequency_to_bin( f19 * 3.0f ) ) ).f == f19 * 3.0f && har.get_MF( 0, 100, Bin( pv.frequency_to_bin( f19 * 3.0f ) ) ).m == pv.get_MF( 0, 100, 19 ).m * 0.25f );
		CHECK( har.get_MF( 0, 100, Bin( pv.frequency_to_bin( f19 * 2.0f ) ) ).m == 0.0f );
		// smear_time with a box distribution: the plain mean of the 2e frames around (PVModify.cpp:580-601); the output starts e frames early
			{
			const Frame e = Frame( p2.time_to_frame( 0.05f ) );
			PV sm = p2.smear_time( 0.05f, 1, 1.0f );
			CHECK( e == 4 && sm.get_num_frames() == F - 1 + 2 * e );
			double mean_m = 0, mean_f = 0;
			for( Frame k = -e; k < e; ++k ) { mean_m += double( p2.get_MF( 1, 30 + k, 77 ).m * 1.0f ); mean_f += double( p2.get_MF( 1, 30 + k, 77 ).f * 1.0f ); }
			CHECK( sm.get_MF( 1, 30 + e, 77 ).m == float( mean_m / ( 2.0 * e ) ) && sm.get_MF( 1, 30 + e, 77 ).f == float( mean_f / ( 2.0 * e ) ) );
			PV sm_default = p2.smear_time( []( TF tf ){ return tf.f < 6000.0f ? 0.03f : 0.08f; } );   // default granularity 5, raised cosine
			CHECK( !sm_default.is_null() && sm_default.get_num_frames() == F - 1 + 2 * Frame( p2.time_to_frame( 0.08f ) ) );
			}
		// modify: a shift by three whole frames and up by four bins lands every MF where it says (rates here make frame <-> time exact:
		// 48000 / 375 = 128); the frequency comes from mod at the MF's own frequency
			{
			PV ex = noise( 1, 30000, 9 ).convert_to_PV( 1024, 375, 1024 );
			const float dt = 3.0f / 128.0f, df = 4.0f * 46.875f;
			PV moved = ex.modify( [=]( TF tf ){ return TF{ tf.t + dt, tf.f + df }; } );
			CHECK( moved.get_num_frames() == ex.get_num_frames() - 1 + 3 );
			CHECK( moved.get_MF( 0, 13, 24 ).m == ex.get_MF( 0, 10, 20 ).m && moved.get_MF( 0, 13, 24 ).f == ex.get_MF( 0, 10, 20 ).f + df );
			CHECK( moved.get_MF( 0, 2, 24 ).m == 0.0f && moved.get_MF( 0, 13, 3 ).m == 0.0f );
			CHECK( ex.modify( []( TF tf ){ return TF{ tf.t + 700.0f, tf.f }; } ).is_null() );      // longer than ten minutes: refused
			CHECK( !ex.modify( []( TF tf ){ return tf; }, Interpolator( []( float v ){ return v; } ) ).is_null() );  // a callable interpolator: sampled table
			}
		// stretch_spline: three output frames per input frame; frame 0 is the input's, the knots are reproduced up to rounding
			{
			PV sp = p2.stretch_spline( 3.0f );
			CHECK( sp.get_num_frames() == 3 * ( F - 1 ) && sp.get_MF( 1, 0, 40 ).m == p2.get_MF( 1, 0, 40 ).m );
			CHECK( close_to( sp.get_MF( 1, 30, 40 ).m, p2.get_MF( 1, 10, 40 ).m, 1e-5 ) && close_to( sp.get_MF( 0, 60, 7 ).f, p2.get_MF( 0, 20, 7 ).f, 1e-5 ) );
			PV sp_var = p2.stretch_spline( [&]( Second t ){ return t < 0.3f ? -5.0f : 2.7f; } );   // below 1 (and negative): one frame per frame
			const Frame early = Frame( std::ceil( 0.3f / p2.frame_to_time( 1 ) ) );
			CHECK( sp_var.get_num_frames() == early + 2 * ( F - 1 - early ) );
			CHECK( p2.cut_frames( 0, 2 ).stretch_spline( 2.0f ).is_null() );            // two frames: the spline needs three
			}
		(void) B;
		}
	// ---- the methods are re-entrant: four host threads run whole chains at once and get what one thread gets
		{
		auto chain = []( uint32_t seed )
			{
			Audio x = noise( 2, 40000 + 1000 * int( seed ), seed );
			PV p = x.convert_to_PV( 2048, 512, 2048 );
			PV q = p.stretch( []( TF tf ){ return 1.0f + tf.t; } ).shape( []( MF mf ){ return MF{ mf.m * 0.5f, mf.f + 10.0f }; } );
			PV r = q.retain_n_loudest_partials( 40 ).add_octaves( []( std::pair<Second, Harmonic> ){ return 0.5f; } );
			return r.convert_to_audio().get_buffer();
			};
		std::vector<std::vector<float>> alone( 4 ), together( 4 );
		for( int t = 0; t < 4; ++t ) alone[size_t( t )] = chain( 100 + uint32_t( t ) );
		std::vector<std::thread> threads;
		for( int t = 0; t < 4; ++t ) threads.emplace_back( [&, t]{ for( int rep = 0; rep < 3; ++rep ) together[size_t( t )] = chain( 100 + uint32_t( t ) ); } );
		for( auto & th : threads ) th.join();
		for( int t = 0; t < 4; ++t )
			CHECK( !alone[size_t( t )].empty() && alone[size_t( t )].size() == together[size_t( t )].size()
				&& std::memcmp( alone[size_t( t )].data(), together[size_t( t )].data(), sizeof( float ) * alone[size_t( t )].size() ) == 0 );
		}
	// ---- const methods of ONE object from several threads (pure reads in the reference): lazy download / upload and the workspace
	//      hand-over are guarded by the object's lock (mirror_lock.h) -- every thread gets the single-threaded answer
		{
		Audio x = noise( 2, 60000, 911 );
		const PV shared_pv = x.convert_to_PV( 2048, 512, 2048 );                       // device resident, pre-pass attached, host copy not yet made
		const std::vector<float> expect = shared_pv.copy().convert_to_audio().get_buffer();
		std::vector<std::vector<float>> got( 6 );
		std::vector<double> energy( 6, 0.0 );
		std::vector<std::thread> threads;
		for( int t = 0; t < 6; ++t ) threads.emplace_back( [&, t]
			{
			if( t & 1 ) { for( const MF & mf : shared_pv.get_buffer() ) energy[size_t( t )] += double( mf.m ); }   // concurrent first download
			got[size_t( t )] = shared_pv.convert_to_audio().get_buffer();            // concurrent hand-over of the one workspace
			} );
		for( auto & th : threads ) th.join();
		for( int t = 0; t < 6; ++t )
			{
			CHECK( got[size_t( t )].size() == expect.size() );
			double worst = 0;
			for( size_t i = 0; i < expect.size() && i < got[size_t( t )].size(); ++i ) worst = std::max( worst, double( std::fabs( got[size_t( t )][i] - expect[i] ) ) );
			CHECK( worst <= 1e-6 );                                                    // (fused and unfused pre-pass differ by rounding of partial sums only)
			}
		CHECK( energy[1] > 0 && energy[1] == energy[3] && energy[3] == energy[5] );
		const Audio shared_audio = noise( 2, 50000, 912 );
		std::vector<Frame> frames( 4, 0 );
		std::vector<std::thread> t2;
		for( int t = 0; t < 4; ++t ) t2.emplace_back( [&, t]{ frames[size_t( t )] = shared_audio.convert_to_PV( 2048, 512, 2048 ).get_num_frames(); } );   // concurrent first upload
		for( auto & th : t2 ) th.join();
		for( int t = 0; t < 4; ++t ) CHECK( frames[size_t( t )] == 50000 / 512 + 1 );
		}
	// ---- cancellation raised from a second thread while a long call is running (flan_CANCEL_POINT, defines.h:49-62): the call comes
	//      back with a null object (or, if it won the race, with the full result) -- never with a partial one
		{
		Audio longish = noise( 8, 48000 * 20, 913 );                                   // a few stages of work: upload, analysis, synthesis
		for( int trial = 0; trial < 4; ++trial )
			{
			std::atomic<bool> flag( false );
			std::thread killer( [&]{ std::this_thread::sleep_for( std::chrono::microseconds( 200 * trial ) ); flag.store( true ); } );
			PV p = longish.convert_to_PV( 2048, 512, 2048, flag );
			Audio back = p.is_null() ? Audio() : p.convert_to_audio( flag );
			killer.join();
			CHECK( p.is_null() || p.get_num_frames() == 48000 * 20 / 512 + 1 );
			CHECK( back.is_null() || back.get_num_frames() == p.get_num_frames() * 512 );
			if( flag.load() && !p.is_null() ) CHECK( p.convert_to_audio( flag ).is_null() );   // once raised, every later call refuses
			}
		}
	// ---- large host <-> device transfers go in slabs through page-locked blocks: what arrives is what was sent
		{
		Audio big_audio = noise( 3, 3000001, 77 );                                    // 36 MB, not a multiple of anything
		const std::vector<float> sent = big_audio.get_buffer();
		const float * d_audio = big_audio.device_data();                               // upload
		std::vector<float> seen( sent.size() );
		CHECK( d_audio != nullptr );
		flanhip_memcpy_d2h( seen.data(), d_audio, sizeof( float ) * seen.size(), nullptr ); flanhip_stream_synchronize( nullptr );
		CHECK( std::memcmp( seen.data(), sent.data(), sizeof( float ) * sent.size() ) == 0 );
		Audio ms_audio = noise( 2, 5000003, 78 ).convert_to_mid_side();               // lives on the device only
		std::vector<float> plain( size_t( 2 ) * 5000003 );
		flanhip_memcpy_d2h( plain.data(), ms_audio.device_data(), sizeof( float ) * plain.size(), nullptr ); flanhip_stream_synchronize( nullptr );
		CHECK( ms_audio.get_buffer().size() == plain.size() && std::memcmp( ms_audio.get_buffer().data(), plain.data(), sizeof( float ) * plain.size() ) == 0 );   // download
		}
	// ---- a grid large enough to go over in slabs (sampled slab k+1 while slab k uploads) = the same grid sampled in one piece
	// and taken through the C ABI by hand; and the block cache hands memory back and forth without mixing results up
		{
		PV big = noise( 1, 700000, 5 ).convert_to_PV( 2048, 512, 2048 );
		const auto g = []( TF tf ){ return 1.0f + 0.35f * std::sin( tf.t * 3.0f ) + tf.f * 1e-5f; };
		CHECK( size_t( big.get_num_frames() ) * big.get_num_bins() * sizeof( float ) >= ( size_t( 4 ) << 20 ) );
		const int64_t F = big.get_num_frames(); const int bins = big.get_num_bins();
		auto grid = big.sample_function_over_domain( Function<TF, float>( g ) );
		void * d_grid = nullptr, * d_max = nullptr, * d_out = nullptr;
		CHECK( flanhip_malloc( &d_grid, sizeof( float ) * grid.size() ) == FLANHIP_OK && flanhip_malloc( &d_max, sizeof( float ) ) == FLANHIP_OK );
		flanhip_memcpy_h2d( d_grid, grid.get_vector().data(), sizeof( float ) * grid.size(), nullptr );
		CHECK( flanhip_stretch_map_dev( static_cast<float*>( d_grid ), F, bins, 48000.0f, 512, static_cast<float*>( d_max ), nullptr ) == FLANHIP_OK );
		float mx = 0; flanhip_memcpy_d2h( &mx, d_max, sizeof( float ), nullptr ); flanhip_stream_synchronize( nullptr );
		const int64_t Fo = int64_t( std::ceil( big.time_to_frame( mx ) ) );
		CHECK( flanhip_malloc( &d_out, sizeof( MF ) * size_t( Fo ) * bins ) == FLANHIP_OK );
		CHECK( flanhip_modify_time_dev( reinterpret_cast<const flanhip_MF*>( big.device_data() ), 1, F, bins, 48000.0f, 512, static_cast<const float*>( d_grid ), Fo,
			static_cast<flanhip_MF*>( d_out ), nullptr ) == FLANHIP_OK );
		std::vector<MF> by_hand( size_t( Fo ) * bins );
		flanhip_memcpy_d2h( by_hand.data(), d_out, sizeof( MF ) * by_hand.size(), nullptr ); flanhip_stream_synchronize( nullptr );
		for( int rep = 0; rep < 3; ++rep )                                            // rep > 0: every block involved comes out of the cache
			{
			PV st = big.stretch( g );
			CHECK( st.get_num_frames() == Fo );
			CHECK( st.get_buffer().size() == by_hand.size() && std::memcmp( st.get_buffer().data(), by_hand.data(), sizeof( MF ) * by_hand.size() ) == 0 );
			Audio out = st.convert_to_audio();
			CHECK( !out.is_null() && out.get_num_frames() == Fo * 512 );
			}
		const float last_time = big.frame_to_time( float( F - 1 ) );
		PV mt_big = big.modify_time( [=]( TF tf ){ return tf.t * 0.5f + ( tf.f > 12000.0f ? 0.25f : 0.0f ); } );   // its length comes from the grid's maximum
		CHECK( mt_big.get_num_frames() == Frame( std::ceil( big.time_to_frame( last_time * 0.5f + 0.25f ) ) ) );
		flanhip_free( d_grid ); flanhip_free( d_max ); flanhip_free( d_out );
		// callables that see the data: the device-only PV goes through the slab pipeline, a PV with a current host copy is read in
		// place; both must give what the device-evaluated affine shaper gives
		CHECK( !big.host_copy_is_current() );
		PV sh_slab = big.shape( []( MF mf ){ return MF{ mf.m * 0.5f, mf.f + 100.0f }; } );
		CHECK( !big.host_copy_is_current() );                                         // shape() did not drag the PV to the host
		PV sh_dev = big.shape_affine( 0.5f, 0.0f, 1.0f, 100.0f );
		CHECK( std::memcmp( sh_slab.get_buffer().data(), sh_dev.get_buffer().data(), sizeof( MF ) * sh_dev.get_buffer().size() ) == 0 );
		PV on_host = big.copy();
		(void) std::as_const( on_host ).get_buffer();
		CHECK( on_host.host_copy_is_current() );
		PV sh_host = on_host.shape( []( MF mf ){ return MF{ mf.m * 0.5f, mf.f + 100.0f }; } );
		CHECK( std::memcmp( sh_host.get_buffer().data(), sh_dev.get_buffer().data(), sizeof( MF ) * sh_dev.get_buffer().size() ) == 0 );
		const auto bend = []( TF tf ){ return tf.f * ( 1.0f + 0.1f * tf.t ) + 20.0f; };
		PV mf_slab = big.modify_frequency( bend ), mf_host = on_host.modify_frequency( bend );
		CHECK( !mf_slab.is_null() && std::memcmp( mf_slab.get_buffer().data(), mf_host.get_buffer().data(), sizeof( MF ) * mf_host.get_buffer().size() ) == 0 );
		}
	}

int main( int argc, char ** argv )
	{
	const bool no_device = argc > 1 && std::string( argv[1] ) == "--no-device";
	if( no_device ) no_device_checks(); else device_checks();
	std::printf( "\n%s (%d failures)\n", failures ? "FAILED" : "PASSED", failures );
	return failures ? 1 : 0;
	}
