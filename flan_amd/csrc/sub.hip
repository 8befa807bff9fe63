// sub.hip -- launchers of the dft 512 / 256 / 128 kernels with several chains per wavefront (pv_kernels_sub.h; Conversions/AudioPV.cpp:12-139).
#include "sub_launch.h"
#include <type_traits>
#include "pv_kernels_sub.h"

namespace flanhip {

#ifndef FLANHIP_SUB_OCC
#define FLANHIP_SUB_OCC 3
#endif
#ifndef FLANHIP_SUB_NV
#define FLANHIP_SUB_NV 4
#endif
static constexpr int kSubWaves = 4, kSubOcc = FLANHIP_SUB_OCC, kSubNv = FLANHIP_SUB_NV;          // wavefronts per block, wavefronts per SIMD the registers are capped for

static int sub_lanes( int dft ) { return dft == 512 ? 32 : dft == 256 ? 16 : dft == 128 ? 8 : 0; }

bool sub_shape( int dft, int W, int hop )
	{
	const int lp = sub_lanes( dft );
	if( !lp || debug_options().force_generic || debug_options().force_direct || debug_options().no_sub ) return false;
	const int step = 2 * lp;
	if( hop % step || W % step || hop > W || W > dft ) return false;
	const int hq = hop / step;
	return hq == 1 || hq == 2 || hq == 4 || hq == 8;
	}

int sub_group_size( int dft ) { return sub_lanes( dft ) ? kSubWaves * ( 64 / sub_lanes( dft ) ) : 0; }

int sub_target_chains( int dft )
	{
	if( const int v = debug_options().target_chains ) { if( v > 0 ) return v; }
	// two wavefronts per SIMD's worth of chains, although the registers are capped for three (no spills at 168): measured (profiles/r06_sub_chains.txt) the
	// longer chains win -- fewer halo frames, a shorter scan -- and a third resident wavefront buys nothing (the kernels are issue- and store-bound)
	return cu_count() * 4 * 2 * ( 64 / sub_lanes( dft ) );
	}

template<int LOG2C, int LP, bool SUMS>
static int launch_analyze_sub( const AnalyzeParams & p, hipStream_t s )
	{
	using L = SubLds<LOG2C, LP>;
	constexpr int NCH = kSubWaves * L::G;
	FLANHIP_REQUIRE( ( int64_t( NCH ) * p.L + 2 ) * std::max( int64_t( p.hop ) * 4, int64_t( ( L::C + 1 ) * 8 ) ) < ( int64_t( 1 ) << 32 ), FLANHIP_ERR_UNSUPPORTED, "chain length x hop too large for the dft 512 / 256 kernels" );
	const size_t lds = L::bytes( kSubWaves, true );
	static_assert( L::bytes( kSubWaves, true ) * kSubOcc <= 160 * 1024, "LDS budget" );
	auto kern = k_analyze_sub<LOG2C, LP, kSubWaves, SUMS, kSubOcc, kSubNv>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + NCH - 1 ) / NCH ) * p.num_channels;
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * kSubWaves ), lds, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int run_analyze_sub( const AnalyzeParams & p, int dft, hipStream_t s )
	{
	FLANHIP_REQUIRE( sub_shape( dft, p.window_size, p.hop ), FLANHIP_ERR_UNSUPPORTED, "not a shape of the dft 512 / 256 kernels" );
	FLANHIP_REQUIRE( p.n >= 2 && p.n < ( int64_t( 1 ) << 31 ) - 8192, FLANHIP_ERR_UNSUPPORTED, "channel length outside the 32-bit sample offsets" );
	if( dft == 512 ) return p.sums ? launch_analyze_sub<8, 32, true>( p, s ) : launch_analyze_sub<8, 32, false>( p, s );
	if( dft == 128 ) return p.sums ? launch_analyze_sub<6, 8, true>( p, s ) : launch_analyze_sub<6, 8, false>( p, s );
	return p.sums ? launch_analyze_sub<7, 16, true>( p, s ) : launch_analyze_sub<7, 16, false>( p, s );
	}

template<int LOG2C, int LP, int HOPQ>
static int launch_synth_sub( const SynthParams & p, hipStream_t s )
	{
	using L = SubLds<LOG2C, LP>;
	constexpr int NCH = kSubWaves * L::G;
	FLANHIP_REQUIRE( ( int64_t( NCH ) * p.L + 2 ) * ( ( L::C + 1 ) * 8 ) < ( int64_t( 1 ) << 32 ), FLANHIP_ERR_UNSUPPORTED, "chain length too large for the dft 512 / 256 kernels" );
	const size_t lds = L::bytes( kSubWaves, false );
	auto kern = k_synthesize_sub<LOG2C, LP, kSubWaves, HOPQ, kSubOcc>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + NCH - 1 ) / NCH ) * p.num_channels;
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * kSubWaves ), lds, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

template<int LOG2C, int LP>
static int run_synth_sub_hop( const SynthParams & p, hipStream_t s )
	{
	switch( p.hop / ( 2 * LP ) )
		{
		case 1: return launch_synth_sub<LOG2C, LP, 1>( p, s );
		case 2: return launch_synth_sub<LOG2C, LP, 2>( p, s );
		case 4: return launch_synth_sub<LOG2C, LP, 4>( p, s );
		case 8: return launch_synth_sub<LOG2C, LP, 8>( p, s );
		}
	return FLANHIP_ERR_UNSUPPORTED;
	}

int run_synth_sub( const SynthParams & p, int dft, hipStream_t s )
	{
	FLANHIP_REQUIRE( sub_shape( dft, p.window_size, p.hop ), FLANHIP_ERR_UNSUPPORTED, "not a shape of the dft 512 / 256 kernels" );
	return dft == 512 ? run_synth_sub_hop<8, 32>( p, s ) : dft == 128 ? run_synth_sub_hop<6, 8>( p, s ) : run_synth_sub_hop<7, 16>( p, s );
	}

} // namespace flanhip
