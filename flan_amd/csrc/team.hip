// team.hip -- launchers of the dft 8192 / 16384 team kernels (pv_kernels_team.h; Conversions/AudioPV.cpp:12-139 at the sizes FFTHelper.cpp:16-26
// hands to FFTW like any other).
#include "team_launch.h"
#include <type_traits>
#include "pv_kernels_team.h"

namespace flanhip {

// ( window / 128 R, hop / 128 R ) the synthesis is instantiated for; hop 0 = HALF a step (64 R samples): with it hop = window / 16 is served at window = dft / 2
// and dft / 4 -- the reference API's default ratio, ( 4096, 256, 8192 ) and ( 8192, 512, 16384 ) -- and ( 16, 1 ) serves it at window = dft
// ... hop -4 / -8 = a QUARTER / an EIGHTH of a step: the API's default hop 128 kept while the sizes grow -- ( 2048, 128, 8192 ) = ( 4, -4 ), ( 2048, 128, 16384 ) =
// ( 2, -8 ), ( 4096, 128, 8192 ) = ( 8, -4 ), ( 4096, 256, 16384 ) = ( 4, -4 ) ...
#define FLANHIP_TEAM_SHAPES( X ) X( 4, 0 ) X( 4, 1 ) X( 4, 2 ) X( 8, 0 ) X( 8, 1 ) X( 8, 2 ) X( 8, 4 ) X( 16, 1 ) X( 16, 2 ) X( 16, 4 ) X( 16, 8 ) \
	X( 2, 0 ) X( 2, 1 ) X( 2, -4 ) X( 2, -8 ) X( 4, -4 ) X( 4, -8 ) X( 8, -4 ) X( 8, -8 ) X( 16, 0 ) X( 16, -4 )

bool team_shape( int dft, int W, int hop )
	{
	const int R = team_radix( dft );
	if( !R || debug_options().force_generic || debug_options().force_direct ) return false;
	const int step = 128 * R;
	if( W % step || hop > W || hop < 1 ) return false;
	const int wq = W / step;
	int hs;                                                                   // whole steps; 0: half a step; -4 / -8: a quarter / an eighth
	if( hop % step == 0 ) hs = hop / step;
	else if( 2 * hop == step ) hs = 0;
	else if( 4 * hop == step ) hs = -4;
	else if( 8 * hop == step ) hs = -8;
	else return false;
#define X( WQ, HS ) if( wq == WQ && hs == HS ) return true;
	FLANHIP_TEAM_SHAPES( X )
#undef X
	return false;
	}

// teams (chains) per block: two teams of four wavefronts share one set of tables and the window; a team of eight has the CU to itself
template<int R> struct TeamsPerBlock { static constexpr int value = R == 4 ? 2 : 1; };
int team_group_size( int dft ) { return team_radix( dft ) == 4 ? 2 : 1; }
int team_target_chains( int dft )
	{
	if( const int v = debug_options().target_chains ) { if( v > 0 ) return v; }
	return cu_count() * team_group_size( dft );
	}

static TeamTables team_tables( const Plan & plan, int R )
	{
	const TeamTableLayout l = team_table_layout( R );
	return TeamTables{ plan.d_team + l.tw1, plan.d_team + l.tw3, plan.d_team + l.twj, plan.d_team + l.tws, plan.d_team + l.two, plan.d_window };
	}

// the window lives in LDS up to 8192 samples (64 R WQ pairs)
template<int R, int WQ> struct TeamWinLds { static constexpr bool value = 128 * R * WQ <= 8192; };

template<int R, int WQ, bool SUMS>
static int launch_analyze_team( const AnalyzeParams & p, const TeamTables & tb, hipStream_t s )
	{
	constexpr int TEAMS = TeamsPerBlock<R>::value;
	constexpr bool WINLDS = TeamWinLds<R, WQ>::value;
	using L = TeamLds<R, TEAMS, WINLDS ? 64 * R * WQ : 0>;
	const size_t lds = L::bytes();
	static_assert( L::bytes() + 64 <= 160 * 1024, "LDS budget" );
	auto kern = k_analyze_team<R, TEAMS, WQ, SUMS, WINLDS>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t chains = int64_t( ( p.chains_per_channel + TEAMS - 1 ) / TEAMS ) * p.num_channels;       // a block = a group of TEAMS chains of one channel
	FLANHIP_REQUIRE( chains < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) chains ), dim3( 64 * R * TEAMS ), lds, s, p, tb );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

template<int R, int WQ, int HS>
static int launch_synth_team( const SynthParams & p, const TeamTables & tb, hipStream_t s )
	{
	constexpr int TEAMS = TeamsPerBlock<R>::value;
	constexpr bool WINLDS = TeamWinLds<R, WQ>::value;
	using L = TeamLds<R, TEAMS, WINLDS ? 64 * R * WQ : 0>;
	const size_t lds = L::bytes();
	static_assert( L::bytes() + 64 <= 160 * 1024, "LDS budget" );
	auto kern = k_synthesize_team<R, TEAMS, WQ, HS, WINLDS>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t chains = int64_t( ( p.chains_per_channel + TEAMS - 1 ) / TEAMS ) * p.num_channels;
	FLANHIP_REQUIRE( chains < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) chains ), dim3( 64 * R * TEAMS ), lds, s, p, tb );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

template<int R>
static int run_analyze_team_r( const AnalyzeParams & p, const TeamTables & tb, hipStream_t s )
	{
	switch( p.window_size / ( 128 * R ) )
		{
		case 2:  return p.sums ? launch_analyze_team<R, 2, true>( p, tb, s ) : launch_analyze_team<R, 2, false>( p, tb, s );
		case 4:  return p.sums ? launch_analyze_team<R, 4, true>( p, tb, s ) : launch_analyze_team<R, 4, false>( p, tb, s );
		case 8:  return p.sums ? launch_analyze_team<R, 8, true>( p, tb, s ) : launch_analyze_team<R, 8, false>( p, tb, s );
		case 16: return p.sums ? launch_analyze_team<R, 16, true>( p, tb, s ) : launch_analyze_team<R, 16, false>( p, tb, s );
		}
	return FLANHIP_ERR_UNSUPPORTED;
	}

int run_analyze_team( const AnalyzeParams & p, const Plan & plan, int dft, hipStream_t s )
	{
	const int R = team_radix( dft );
	FLANHIP_REQUIRE( R && plan.d_team && team_shape( dft, p.window_size, p.hop ), FLANHIP_ERR_UNSUPPORTED, "not a team shape" );
	FLANHIP_REQUIRE( p.n < ( int64_t( 1 ) << 31 ) - 65536, FLANHIP_ERR_UNSUPPORTED, "channel too long for 32-bit sample offsets" );
	const TeamTables tb = team_tables( plan, R );
	return R == 4 ? run_analyze_team_r<4>( p, tb, s ) : run_analyze_team_r<8>( p, tb, s );
	}

template<int R>
static int run_synth_team_r( const SynthParams & p, const TeamTables & tb, hipStream_t s )
	{
	const int step = 128 * R, wq = p.window_size / step;
	const int hs = p.hop % step == 0 ? p.hop / step : 2 * p.hop == step ? 0 : 4 * p.hop == step ? -4 : -8;      // (team_shape has admitted nothing else)
#define X( WQ, HS ) if( wq == WQ && hs == HS ) return launch_synth_team<R, WQ, HS>( p, tb, s );
	FLANHIP_TEAM_SHAPES( X )
#undef X
	return FLANHIP_ERR_UNSUPPORTED;
	}

int run_synth_team( const SynthParams & p, const Plan & plan, int dft, hipStream_t s )
	{
	const int R = team_radix( dft );
	FLANHIP_REQUIRE( R && plan.d_team && team_shape( dft, p.window_size, p.hop ), FLANHIP_ERR_UNSUPPORTED, "not a team shape" );
	const TeamTables tb = team_tables( plan, R );
	return R == 4 ? run_synth_team_r<4>( p, tb, s ) : run_synth_team_r<8>( p, tb, s );
	}

} // namespace flanhip
