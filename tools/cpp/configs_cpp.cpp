// configs_cpp.cpp -- BASELINE configs 3 and 5 through the C++ drop-in surface (include/flan/*.h), end to end and per call:
//   3: 8 ch x 60 s noise -> convert_to_PV(2048,512,2048) -> stretch( lambda -> 2.0f ) -> convert_to_audio
//   5: 2 ch x 60 s at 96 kHz -> resample(48000) -> convert_to_PV -> shape( lambda: f + 100 ) -> convert_to_audio
// What the calls cost INCLUDING the host work the reference's API implies (the factor is a host callable: it is sampled on the
// host over the (frame, bin) grid exactly like Function.h:155-171 does, then uploaded).  Build: see tools/cpp/Makefile.
#include <chrono>
#include <cstdio>
#include <vector>

#include "flan/flan.h"

using namespace flan;
using clk = std::chrono::steady_clock;
static double ms( clk::time_point a, clk::time_point b ) { return std::chrono::duration<double, std::milli>( b - a ).count(); }
static uint32_t hash32( uint32_t x ) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

int main()
	{
	const int ch = 8, n = 60 * 48000;
	std::vector<float> x( size_t( ch ) * n );
	for( size_t i = 0; i < x.size(); ++i ) x[i] = float( hash32( uint32_t( i ) * 2654435761u ) >> 8 ) * ( 1.0f / 8388608.0f ) - 1.0f;
	Audio a = Audio::create_from_buffer( std::move( x ), ch, 48000.0f );
	std::printf( "host workers: %d\n", detail::host_workers() );
	for( int rep = 0; rep < 6; ++rep )                                             // rep 0 warms the device and uploads the audio
		{
		const auto t0 = clk::now();
		PV pv = a.convert_to_PV( 2048, 512, 2048 );
		const auto t1 = clk::now();
		PV st_l = pv.stretch( []( TF ){ return 2.0f; } );                            // config 3 as written: a callable
		const auto t2 = clk::now();
		Audio out_l = st_l.convert_to_audio();
		const auto t3 = clk::now();
		PV st_c = pv.stretch( 2.0f );                                                // the same with a constant: nothing sampled on the host
		const auto t4 = clk::now();
		Audio out_c = st_c.convert_to_audio();
		const auto t5 = clk::now();
		if( pv.is_null() || out_l.is_null() || out_c.is_null() ) { std::printf( "FAILED\n" ); return 1; }
		if( rep >= 4 )                                                              // what reading the result on the host costs on top
			{
			const auto t6 = clk::now();
			const std::vector<float> & samples = out_c.get_buffer();
			const auto t7 = clk::now();
			std::printf( "        out.get_buffer(): %zu samples to the host in %.3f ms (%.1f GB/s)\n", samples.size(), ms( t6, t7 ), samples.size() * 4e-6 / ms( t6, t7 ) );
			}
		std::printf( "rep %d: convert_to_PV %.3f ms | stretch(lambda) %.3f ms, convert_to_audio %.3f ms | stretch(2.0f) %.3f ms, convert_to_audio %.3f ms"
			" | config 3 end to end: %.3f ms (lambda) / %.3f ms (constant)\n", rep, ms( t0, t1 ), ms( t1, t2 ), ms( t2, t3 ), ms( t3, t4 ), ms( t4, t5 ),
			ms( t0, t3 ), ms( t0, t1 ) + ms( t3, t5 ) );
		}
	// config 5: the shaper is a host callable that sees every MF (PV.cpp:435-436): the PV comes to the host, the callable runs on
	// every bin, the shaped table goes back; shape_affine (extension) evaluates the same affine shaper on the device
		{
		const int ch5 = 2, n5 = 60 * 96000;
		std::vector<float> x5( size_t( ch5 ) * n5 );
		for( size_t i = 0; i < x5.size(); ++i ) x5[i] = float( hash32( uint32_t( i ) * 2246822519u ) >> 8 ) * ( 1.0f / 8388608.0f ) - 1.0f;
		Audio a5 = Audio::create_from_buffer( std::move( x5 ), ch5, 96000.0f );
		for( int rep = 0; rep < 6; ++rep )
			{
			const auto t0 = clk::now();
			Audio r = a5.resample( 48000.0f );
			const auto t1 = clk::now();
			PV p5 = r.convert_to_PV( 2048, 512, 2048 );
			const auto t2 = clk::now();
			PV sh = p5.shape( []( MF mf ){ return MF{ mf.m, mf.f + 100.0f }; } );
			const auto t3 = clk::now();
			Audio o5 = sh.convert_to_audio();
			const auto t4 = clk::now();
			PV sh_a = p5.shape_affine( 1.0f, 0.0f, 1.0f, 100.0f );
			const auto t5 = clk::now();
			if( r.is_null() || sh.is_null() || o5.is_null() || sh_a.is_null() ) { std::printf( "FAILED\n" ); return 1; }
			std::printf( "config 5 rep %d: resample %.3f ms | convert_to_PV %.3f ms | shape(lambda) %.3f ms | convert_to_audio %.3f ms | end to end %.3f ms"
				" | shape_affine instead: %.3f ms\n", rep, ms( t0, t1 ), ms( t1, t2 ), ms( t2, t3 ), ms( t3, t4 ), ms( t0, t4 ), ms( t4, t5 ) );
			}
		}
	PV pv = a.convert_to_PV( 2048, 512, 2048 );
	Function<TF, float> two( []( TF ){ return 2.0f; } );
	for( int rep = 0; rep < 8; ++rep )                                              // the host part of stretch( callable ) on its own
		{
		const auto t0 = clk::now();
		auto grid = pv.sample_function_over_domain( two );
		const auto t1 = clk::now();
		std::printf( "sampling %zu points: %.3f ms\n", grid.size(), ms( t0, t1 ) );
		}
	return 0;
	}
