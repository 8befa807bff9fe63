// conversions.hip -- Audio::convert_to_PV and PV::convert_to_audio behind the C ABI
// (reference: Conversions/AudioPV.cpp:12-78 and :86-139).
#include "flanhip_internal.h"
#include <atomic>
#include <memory>
#include "pv_kernels.h"
#include "pv_kernels_fast.h"
#include <type_traits>
#include "pv_kernels_v2.h"
#include "pv_kernels_v3.h"
#include "pv_kernels_eo.h"
#include "pv_kernels_any.h"
#include "pv_kernels_mr.h"
#include "pv_kernels_bs.h"
#include "pv_kernels_big.h"
#include "team_launch.h"
#include "sub_launch.h"
#include <algorithm>
#include <cstdlib>

namespace flanhip {

// dft 2048 has ONE kernel generation (pv_kernels_v2.h); flanhip_debug_option's ANA / SYN_VARIANT select the phase-ablated instantiations of
// diagnostic builds (FLANHIP_ABLATIONS: 101 ... for the analysis, 102 ... for the synthesis) and, for the synthesis, 2 = behind the scan
// kernel even where it could work out its own carries.  The A/B predecessor of EVERY tuned size is the generic kernel pair of pv_kernels.h
// (FORCE_GENERIC; at dft 4096 also per kernel: ANA4096_OLD / SYN4096_OLD) -- round 1's own tuned kernels for dft 2048 / 4096 (k_analyze_fast,
// k_synthesize_fast) and the spare configurations of the dft 1024 / 512 kernels were retired in round 6.  All per calling thread (core.hip: debug_options).

static constexpr size_t kMaxLds = 160 * 1024;   // gfx950: 160 KiB LDS per CU, one workgroup may take all of it

// dft sizes: powers of two in [32, 8192] have FFT kernels (tuned or LDS-resident); of the other EVEN sizes the reference would hand to FFTW
// (FFTHelper.cpp:16-26) the mixed-radix, chirp-z and residue-pair kernels serve what mr_size / bs_size / big_size below say, and the direct-sum
// kernels of pv_kernels_any.h the rest.  Odd sizes are refused: the reference's own
// PVBuffer derives the dft size as ( bins - 1 ) * 2 (PVBuffer.cpp:356-359), so an odd one does not survive its own round trip.
static constexpr int kMaxAnyDft = 1 << 20;
static bool fft_size( int dft ) { return is_pow2( dft ) && dft >= 32 && dft <= 8192; }
// ... and the sizes the mixed-radix kernels serve (pv_kernels_mr.h): half the size a product of 2, 3, 5, 7, 11, 13, at most 8192, and the block's
// tables and state within the LDS of a CU for this window
static int mr_blocks_per_cu( const MrPlan & pl, int W )
	{
	const size_t lds = std::max( mr_analyze_lds( pl.C, W, pl.win_lds, pl.kc_lds ), mr_synth_lds( pl.C, W, pl.win_lds, pl.kc_lds, pl.ring_ws != 0 ) );
	if( lds > kMaxLds ) return 0;
	return int( std::min<size_t>( mr_pingpong( pl.C ) ? 2 : 1, kMaxLds / lds ) );      // (blocks of 8 wavefronts; four / two wavefronts per SIMD: the kernels' register budgets)
	}
static bool mr_size( int dft, int W, MrPlan * pl = nullptr )
	{
	MrPlan best;
	if( fft_size( dft ) || debug_options().force_direct || !mr_make_plan( dft, &best ) ) return false;
	// where the optional tables go: the layout that keeps the most blocks resident per CU; between equals the one with more tables in LDS
	int best_blocks = 0;
	for( int kc = 0; kc < 2; ++kc ) for( int wl = 0; wl < 2; ++wl )
		{
		MrPlan t = best; t.kc_lds = kc; t.win_lds = wl; t.ring_ws = 0;
		const int b = mr_blocks_per_cu( t, W );
		if( b > best_blocks || ( b == best_blocks && b > 0 && kc + wl >= best.kc_lds + best.win_lds ) ) { best = t; best_blocks = b; }
		}
	if( best_blocks == 0 )
		{
		// the synthesis' ring does not fit beside the transform and the running phases (windows above ~6000 samples at dft 16384): the ring in the workspace
		MrPlan t = best; t.kc_lds = 0; t.win_lds = 0; t.ring_ws = 1;
		best_blocks = mr_blocks_per_cu( t, W );
		if( best_blocks > 0 ) best = t;
		}
	if( best_blocks == 0 ) return false;
	if( pl ) *pl = best;
	return true;
	}
// chains the device holds at once for them: one block per chain
static int mr_target_chains( int dft, int W )
	{
	MrPlan pl{};
	if( !mr_size( dft, W, &pl ) ) return cu_count();
	return cu_count() * std::max( 1, mr_blocks_per_cu( pl, W ) );
	}

// ... and the sizes the chirp-z kernels serve (pv_kernels_bs.h, bs_plan.h): half the size with a prime factor above 13, 64 <= C <= 4096, the
// block's tables and state within the LDS of a CU for this window
// the ping-pong layout's kernels with the per-thread table values in registers (pv_kernels_bs.h HOIST: 256 registers, so one block per CU) where
// the LDS holds one block per CU anyway (M = 4096: (2048, 512, 2998) 4.46 -> 3.94 ms; at M = 2048 / 1024 the second block is worth more: 4.07 ms
// against 5.8 at dft 2018).  The diagnostic hook ANA_VARIANT = 1 picks the kernels that read the tables every frame everywhere.
static size_t bs_lds( const BsPlan & pl, int W ) { return std::max( bs_analyze_lds( pl.C, pl.M ), bs_synth_lds( pl.C, pl.M, W, pl.win_lds != 0 ) ); }
static bool bs_hoist( const BsPlan & pl, int W )
	{
	return bs_pingpong( pl.M ) && pl.C < 4 * MR_THREADS && pl.M / 8 <= MR_THREADS && 2 * bs_lds( pl, W ) > kMaxLds && debug_options().ana_variant != 1;
	}
static int bs_blocks_per_cu( const BsPlan & pl, int W )
	{
	if( pl.glob ) return 2;                                                     // (no LDS: buffers and state in device memory; two blocks' stretches of it per CU stay L2-sized)
	const size_t lds = bs_lds( pl, W );
	if( lds > kMaxLds ) return 0;
	return int( std::min<size_t>( bs_pingpong( pl.M ) ? 2 : 1, kMaxLds / lds ) );
	}
static bool bs_size( int dft, int W, BsPlan * pl = nullptr )
	{
	BsPlan best;
	if( fft_size( dft ) || debug_options().force_direct || !bs_plan_in_use( dft, &best ) ) return false;
	if( best.glob )
		{
		// M above 8192 (round 6): for what nothing else serves (bs_plan_in_use: the residue-pair kernels come first where their plan exists)
		best.win_lds = 0;
		if( pl ) *pl = best;
		return true;
		}
	int best_blocks = 0;
	for( int wl = 0; wl < 2; ++wl )
		{
		BsPlan t = best; t.win_lds = wl;
		const int b = bs_blocks_per_cu( t, W );
		if( b > 0 && b >= best_blocks ) { best = t; best_blocks = b; }
		}
	if( best_blocks == 0 ) return false;
	if( pl ) *pl = best;
	return true;
	}
static int bs_target_chains( int dft, int W )
	{
	BsPlan pl{};
	if( !bs_size( dft, W, &pl ) ) return cu_count();
	return cu_count() * std::max( 1, bs_blocks_per_cu( pl, W ) );
	}

// ... and the sizes above 16384 the residue-pair kernels serve (pv_kernels_big.h): half the size = C1 x C2 with C2 = 1024 ... 4096 a power of two, or
// (round 6) any product of 2 ... 13 between 256 and 4096, C1 <= 256 (bs_plan.h: big_make_plan)
static bool big_size( int dft, int W, BigPlan * pl = nullptr )
	{
	BigPlan t;
	if( fft_size( dft ) || debug_options().force_direct || dft <= 16384 || !big_make_plan( dft, W, &t ) ) return false;
	// (the synthesis' overlap-add ring of W floats: in LDS where it fits beside the two transforms, in the workspace otherwise -- big_ring_in_workspace)
	if( pl ) *pl = t;
	return true;
	}
static bool big_ring_in_workspace( const BigPlan & pl, int W ) { return big_synth_lds( pl.C2, W ) > kMaxLds; }
static int big_target_chains() { return cu_count(); }        // one block per CU, a chain is P blocks: P rounds

static bool dft_size_ok( int dft ) { return dft >= 4 && dft % 2 == 0 && dft <= kMaxAnyDft; }

// unit[m] = ( cos, sin )( 2 pi m / N ) in double, exact at the quarter turns, for the direct-sum kernels; per ( device, N ), the four most
// recently used sizes (16 bytes x N each, up to 2^20: a sweep over sizes must not pile them up), handed out as shared references like the plans
struct UnitRef { d2 * d = nullptr; UnitRef() = default; UnitRef( const UnitRef & ) = delete; UnitRef & operator=( const UnitRef & ) = delete; ~UnitRef() { (void) hipFree( d ); (void) hipGetLastError(); } };
static std::mutex g_unit_mutex;
static std::vector<std::pair<std::pair<int, int>, std::shared_ptr<const UnitRef>>> & unit_cache() { static auto * v = new std::vector<std::pair<std::pair<int, int>, std::shared_ptr<const UnitRef>>>; return *v; }
static int get_unit_circle( int N, std::shared_ptr<const UnitRef> * out )
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	const auto key = std::make_pair( device, N );
	auto lookup = [&]() -> bool
		{
		auto & c = unit_cache();
		for( size_t i = 0; i < c.size(); ++i ) if( c[i].first == key ) { auto e = c[i]; c.erase( c.begin() + long( i ) ); c.push_back( e ); *out = e.second; return true; }
		return false;
		};
		{
		std::lock_guard<std::mutex> lock( g_unit_mutex );
		if( lookup() ) return FLANHIP_OK;
		}
	std::vector<d2> u( size_t( N ), d2{ 1.0, 0.0 } );                               // (outside the lock)
	const long double two_pi = 6.283185307179586476925286766559005768L;
	for( int m = 0; m < N; ++m ) u[size_t( m )] = d2{ double( cosl( two_pi * m / N ) ), double( sinl( two_pi * m / N ) ) };
	u[0] = d2{ 1.0, 0.0 }; u[size_t( N / 2 )] = d2{ -1.0, 0.0 };
	if( N % 4 == 0 ) { u[size_t( N / 4 )] = d2{ 0.0, 1.0 }; u[size_t( 3 * ( N / 4 ) )] = d2{ 0.0, -1.0 }; }
	auto ref = std::make_shared<UnitRef>();
	FLANHIP_CHECK( hipMalloc( &ref->d, sizeof( d2 ) * size_t( N ) ) );
	FLANHIP_CHECK( hipMemcpy( ref->d, u.data(), sizeof( d2 ) * size_t( N ), hipMemcpyHostToDevice ) );
	std::shared_ptr<const UnitRef> evicted;                                         // (released outside the lock: freeing waits for the device)
		{
		std::lock_guard<std::mutex> lock( g_unit_mutex );
		if( lookup() ) return FLANHIP_OK;
		auto & c = unit_cache();
		c.emplace_back( key, ref );
		size_t mine = 0;                                                            // (four per DEVICE, like the plans' bound)
		for( const auto & e : c ) mine += e.first.first == device;
		if( mine > 4 ) for( size_t i = 0; i < c.size(); ++i ) if( c[i].first.first == device ) { evicted = c[i].second; c.erase( c.begin() + long( i ) ); break; }
		}
	*out = ref;
	return FLANHIP_OK;
	}
// chains for the direct-sum kernels: enough blocks ( chains x bin blocks ) to fill the chip, chains of at least 7 frames (the halo frame
// every chain but a channel's first recomputes is then <= 1/8 of its work)
static int any_target_chains( int bins ) { return std::max( 16, 4096 / ( ( bins + ANY_THREADS - 1 ) / ANY_THREADS ) ); }

// chains: the chains a block walks (one per wavefront; or ONE, walked by a team of several wavefronts, see k_analyze)
static size_t analyze_lds_bytes( int C, int W, int chains, bool state_in_lds )
	{
	const int wpad = ( W + 3 ) & ~3;
	if( state_in_lds ) return size_t( wpad ) * 4 + size_t( chains ) * ( size_t( padded_len( C ) ) * 8 + size_t( C + 4 ) * 4 );   // k_analyze BIG: no twiddles, previous phases
	return size_t( C ) * 8 + size_t( wpad ) * 4 + size_t( chains ) * padded_len( C ) * 8;
	}

static size_t synth_lds_bytes( int C, int W, int chains, bool state_in_lds )
	{
	const int wpad = ( W + 3 ) & ~3;
	if( state_in_lds ) return size_t( wpad ) * 4 + size_t( chains ) * ( size_t( padded_len( C + 1 ) ) * 8 + size_t( wpad ) * 4 + size_t( C + 2 ) * 8 );   // k_synthesize BIG
	return size_t( C ) * 8 + size_t( wpad ) * 4 + size_t( chains ) * ( size_t( padded_len( C + 1 ) ) * 8 + size_t( wpad ) * 4 );
	}

template<int LOG2C, int WAVES, int T = 1>
static int run_analyze( const AnalyzeParams & p, hipStream_t s )
	{
	const size_t lds = analyze_lds_bytes( 1 << LOG2C, p.window_size, WAVES, LOG2C >= 12 && T == 1 );
	FLANHIP_REQUIRE( lds <= kMaxLds, FLANHIP_ERR_UNSUPPORTED, "window/dft too large for LDS" );
	auto kern = k_analyze<LOG2C, WAVES, T>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t chains = int64_t( p.chains_per_channel ) * p.num_channels;
	const int64_t blocks = ( chains + WAVES - 1 ) / WAVES;
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * WAVES * T ), lds, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

// Tuned kernels.  dft 2048: 16 complex points per lane, <= 256 VGPRs -> 2 wavefronts per SIMD: one 8-wave block per CU
// (LDS 94 KB; 164 KB with the fused sums), 2048 resident chains.  dft 4096: teams of two such wavefronts (pv_kernels_eo.h), 1024 resident chains.
// dft 1024 / 512 (pv_kernels_v3.h): { index, wavefronts per block = chains per group, wavefronts per SIMD the registers are capped for, bins per vector
// stream, frames the sample requests run ahead }.  One configuration per size since round 6 (the occupancy experiment of round 5 ran three: dft 1024 is as
// fast at two, three and four wavefronts per SIMD in the analysis and 10 % faster at four in the synthesis -- profiles/r05_v3_variants.txt; its partners
// ( 4, 3, 4 ), ( 8, 2, 8 ) at dft 1024 and ( 8, 4, 4 ), ( 8, 2, 4 ) at dft 512 are in the history, not in the library).  dft 512 on the grid of
// pv_kernels_sub.h runs those kernels; this configuration serves its other shapes and FLANHIP_DEBUG_NO_SUB.
#define FLANHIP_V3_CFGS_9( X ) X( 0, 8, 4, 4, 1 )
#define FLANHIP_V3_CFGS_8( X ) X( 0, 4, 3, 4, 1 )
struct V3Cfg { int waves, occ, nv; };
static V3Cfg v3_cfg( int dft )
	{
	const int v = debug_options().ana_variant;
#define X( I, W, O, N, P ) if( v == I ) return V3Cfg{ W, O, N };
	if( dft == 1024 ) { FLANHIP_V3_CFGS_9( X ) return V3Cfg{ 8, 4, 4 }; }
	FLANHIP_V3_CFGS_8( X )
#undef X
	return V3Cfg{ 4, 3, 4 };
	}
static int v3_index( int dft )
	{
	const int v = debug_options().ana_variant;
#define X( I, W, O, N, P ) if( v == I ) return I;
	if( dft == 1024 ) { FLANHIP_V3_CFGS_9( X ) return 0; }
	FLANHIP_V3_CFGS_8( X )
#undef X
	return 0;
	}
static bool v3_size( int dft ) { return dft == 1024 || dft == 512; }
// chains per group = wavefronts per block of the kernels that pass group totals from the analysis to the synthesis (0: none at this size)
static int group_size_of( int dft ) { return dft == 2048 ? 8 : dft == 4096 ? 4 : v3_size( dft ) ? v3_cfg( dft ).waves : 8; }
static constexpr int kTeamWaves12 = 8;           // generic kernels at dft 8192: one chain per block of 8 wavefronts (8 bins per thread), one block per CU
// chains the chip holds at once for the generic kernels (one chain per team from dft 1024 up, LDS decides how many teams a CU takes)
static int generic_target_chains( int dft ) { const int cus = cu_count(); return dft >= 8192 ? cus : dft == 4096 ? 2 * cus : dft == 2048 ? 4 * cus : 16 * cus; }
// ... of the tuned kernels: every SIMD of the device holds two wavefronts of the dft 2048 kernels (8 chains per CU), one two-wavefront team
// of the dft 4096 ones (4 per CU) -- counted from the device's own CU count (a CPX partition has 32, not 256)
static int fast_target_chains( int dft, bool synth )
	{
	if( const int v = debug_options().target_chains ) { if( v > 0 ) return v; }
	// dft 1024 / 512 (pv_kernels_v3.h): four / eight wavefronts per SIMD
	return cu_count() * ( dft == 4096 ? 4 : dft == 2048 ? 8 : 4 * v3_cfg( dft ).occ );
	}

template<int WAVES, bool SUMS, int NV, int ABL = 0>
static int run_analyze_v2( const AnalyzeParams & p, const FastTables & tb, hipStream_t s )
	{
	// 32-bit byte offsets from the block's first halo frame: ( WAVES L + 1 ) frames of hop samples / of 1025 MFs
	FLANHIP_REQUIRE( ( int64_t( WAVES ) * p.L + 2 ) * std::max( int64_t( p.hop ) * 4, int64_t( 8200 ) ) < ( int64_t( 1 ) << 32 ), FLANHIP_ERR_UNSUPPORTED, "chain length x hop too large for the dft 2048 kernel" );
	const size_t lds = V2Lds::bytes( WAVES );
	static_assert( V2Lds::bytes( WAVES ) <= kMaxLds, "LDS budget" );
	auto kern = k_analyze_v2<WAVES, SUMS, NV, ABL>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + WAVES - 1 ) / WAVES ) * p.num_channels;    // a block = a group of WAVES chains of one channel
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * WAVES ), lds, s, p, tb );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

template<bool SUMS>
static int run_analyze_v2_variant( int v, const AnalyzeParams & p, const FastTables & tb, hipStream_t s )
	{
	switch( v )
		{
		default: return run_analyze_v2<8, SUMS, 8>( p, tb, s );                    // 8 bins at a time (16 and 4 were measured and dropped)
#ifdef FLANHIP_ABLATIONS
		case 101: return run_analyze_v2<8, SUMS, 8, 1>( p, tb, s );
		case 102: return run_analyze_v2<8, SUMS, 8, 2>( p, tb, s );
		case 104: return run_analyze_v2<8, SUMS, 8, 4>( p, tb, s );
		case 108: return run_analyze_v2<8, SUMS, 8, 8>( p, tb, s );
		case 116: return run_analyze_v2<8, SUMS, 8, 16>( p, tb, s );
		case 132: return run_analyze_v2<8, SUMS, 8, 32>( p, tb, s );
		case 106: return run_analyze_v2<8, SUMS, 8, 6>( p, tb, s );
		case 124: return run_analyze_v2<8, SUMS, 8, 24>( p, tb, s );
		case 163: return run_analyze_v2<8, SUMS, 8, 63>( p, tb, s );
		case 164: return run_analyze_v2<8, SUMS, 8, 64>( p, tb, s );
		case 228: return run_analyze_v2<8, SUMS, 8, 128>( p, tb, s );
		case 292: return run_analyze_v2<8, SUMS, 8, 192>( p, tb, s );
		case 139: return run_analyze_v2<8, SUMS, 8, 39>( p, tb, s );
		case 107: return run_analyze_v2<8, SUMS, 8, 7>( p, tb, s );
		case 356: return run_analyze_v2<8, SUMS, 8, 256>( p, tb, s );
		case 1124: return run_analyze_v2<8, SUMS, 8, 1024>( p, tb, s );
		case 1163: return run_analyze_v2<8, SUMS, 8, 1024 + 39>( p, tb, s );
		case 612: return run_analyze_v2<8, SUMS, 8, 512>( p, tb, s );
		case 868: return run_analyze_v2<8, SUMS, 8, 768>( p, tb, s );
#endif
		}
	}

// dft 1024 / 512: one wavefront per chain, 8 / 4 points per lane (pv_kernels_v3.h)
template<int LOG2C, bool SUMS, int WAVES, int OCC, int NV, int PF>
static int run_analyze_v3( const AnalyzeParams & p, hipStream_t s )
	{
	using L = V3Lds<LOG2C>;
	FLANHIP_REQUIRE( ( int64_t( WAVES ) * p.L + 2 ) * std::max( int64_t( p.hop ) * 4, int64_t( ( L::C + 1 ) * 8 ) ) < ( int64_t( 1 ) << 32 ), FLANHIP_ERR_UNSUPPORTED, "chain length x hop too large for the dft 1024 / 512 kernel" );
	const size_t lds = L::bytes( WAVES, true );
	static_assert( L::bytes( WAVES, true ) * ( 4 * OCC / WAVES ) <= kMaxLds, "LDS budget" );
	auto kern = k_analyze_v3<LOG2C, WAVES, SUMS, OCC, NV, ( PF & 255 ), ( PF >> 8 )>;    // (diagnostic configurations carry an ablation mask in the high bits)
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + WAVES - 1 ) / WAVES ) * p.num_channels;    // a block = a group of WAVES chains of one channel
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * WAVES ), lds, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}
template<bool SUMS>
static int run_analyze_v3_cfg( int dft, const AnalyzeParams & p, hipStream_t s )
	{
	const int idx = v3_index( dft );
#define X( I, W, O, N, P ) if( idx == I ) return run_analyze_v3<9, SUMS, W, O, N, P>( p, s );
	if( dft == 1024 ) { FLANHIP_V3_CFGS_9( X ) }
#undef X
#define X( I, W, O, N, P ) if( idx == I ) return run_analyze_v3<8, SUMS, W, O, N, P>( p, s );
	if( dft == 512 ) { FLANHIP_V3_CFGS_8( X ) }
#undef X
	return FLANHIP_ERR_UNSUPPORTED;
	}

template<int LOG2C, int HOPQ, int WAVES, int OCC>
static int run_synth_v3( const SynthParams & p, hipStream_t s )
	{
	using L = V3Lds<LOG2C>;
	FLANHIP_REQUIRE( ( int64_t( WAVES ) * p.L + 2 ) * ( ( L::C + 1 ) * 8 ) < ( int64_t( 1 ) << 32 ), FLANHIP_ERR_UNSUPPORTED, "chain length too large for the dft 1024 / 512 kernel" );
	const size_t lds = L::bytes( WAVES, false ) + ( HOPQ == 0 ? size_t( WAVES ) * size_t( ( p.window_size + 63 ) & ~63 ) * 4 : 0 );   // (HOPQ 0: a ring of `window` floats per wavefront behind the buffers)
	auto kern = k_synthesize_v3<LOG2C, WAVES, HOPQ, OCC>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + WAVES - 1 ) / WAVES ) * p.num_channels;
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * WAVES ), lds, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}
static int synth_fast_kind( int dft, int W, int hop );
template<int LOG2C, int WAVES, int OCC>
static int run_synth_v3_hopq( const SynthParams & p, hipStream_t s )
	{
	if( synth_fast_kind( 2 << LOG2C, p.window_size, p.hop ) == 2 ) return run_synth_v3<LOG2C, 0, WAVES, OCC>( p, s );
	switch( p.hop / 128 )
		{
		case 1: return run_synth_v3<LOG2C, 1, WAVES, OCC>( p, s );
		case 2: return run_synth_v3<LOG2C, 2, WAVES, OCC>( p, s );
		case 4: return run_synth_v3<LOG2C, 4, WAVES, OCC>( p, s );
		case 8: if constexpr( LOG2C == 9 ) return run_synth_v3<LOG2C, 8, WAVES, OCC>( p, s );
		}
	return FLANHIP_ERR_UNSUPPORTED;
	}
static int run_synth_v3_hop( int dft, const SynthParams & p, hipStream_t s )
	{
	const int idx = v3_index( dft );
	// (the synthesis kernels have no NV: configurations that differ in it alone share an instantiation)
#define X( I, W, O, N, P ) if( idx == I ) return run_synth_v3_hopq<9, W, O>( p, s );
	if( dft == 1024 ) { FLANHIP_V3_CFGS_9( X ) }
#undef X
#define X( I, W, O, N, P ) if( idx == I ) return run_synth_v3_hopq<8, W, O>( p, s );
	if( dft == 512 ) { FLANHIP_V3_CFGS_8( X ) }
#undef X
	return FLANHIP_ERR_UNSUPPORTED;
	}

// dft 4096 with window <= 2048 as two 1024-point register transforms per frame (pv_kernels_eo.h): teams of two wavefronts, 160 KB of LDS
template<int TEAMS, bool SUMS, int QV, bool DOUBLE = false, bool WBIG = false>
static int run_analyze_eo_team( const AnalyzeParams & p, const FastTables & tb, hipStream_t s )
	{
	using L = typename std::conditional<WBIG, EoLdsBig, EoLds>::type;
	const size_t lds = L::bytes( DOUBLE ? 2 * TEAMS : TEAMS );                 // a team's two wavefronts share one E and one O buffer (or two of each)
	static_assert( L::bytes( DOUBLE ? 2 * TEAMS : TEAMS ) <= kMaxLds, "LDS budget" );
	auto kern = k_analyze_eo_team<TEAMS, SUMS, QV, DOUBLE, WBIG>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + TEAMS - 1 ) / TEAMS ) * p.num_channels;   // a block = a group of TEAMS chains of one channel
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 128 * TEAMS ), lds, s, p, tb );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

// Which tuned synthesis kernel serves this shape?  0: none (generic kernels); 1: overlap-add accumulator in registers (hop 128 /
// 256 / 512 / 1024, window a multiple of 128); 2: accumulator as an LDS ring (any hop <= window, any window <= dft).
static int synth_fast_kind( int dft, int W, int hop )
	{
	if( !( dft == 2048 || dft == 4096 || v3_size( dft ) ) || hop > W || hop < 1 || force_generic() ) return 0;
	const int hq = hop / 128;
	if( v3_size( dft ) ) return ( hop % 128 == 0 && ( hq == 1 || hq == 2 || hq == 4 || ( hq == 8 && dft == 1024 ) ) && W % 128 == 0 ) ? 1 : 2;   // (2: k_synthesize_v3's LDS-ring form, round 5)
	if( dft == 4096 )
		{
		// the team kernels (pv_kernels_eo.h): accumulator in registers on the grid (hop 128 / 256 / 512 / 1024, windows that are multiples of 256), as an LDS ring
		// otherwise; FLANHIP_DEBUG_SYN4096_OLD: the generic kernels (pv_kernels.h), the A/B predecessor
		if( debug_options().syn11_old ) return 0;
		return ( W % 256 == 0 && ( hop == 128 || hop == 256 || hop == 512 || hop == 1024 ) ) ? 1 : 2;
		}
	if( hop % 128 == 0 && ( hq == 1 || hq == 2 || hq == 4 || hq == 8 ) && W % 128 == 0 ) return 1;
	return 2;
	}

template<int WAVES, int HOPQ, int ABL = 0>
static int run_synth_v2( const SynthParams & p, const FastTables & tb, hipStream_t s )
	{
	FLANHIP_REQUIRE( ( int64_t( WAVES ) * p.L + 2 ) * 8200 < ( int64_t( 1 ) << 32 ), FLANHIP_ERR_UNSUPPORTED, "chain length too large for the dft 2048 kernel" );   // 32-bit byte offsets inside a block's frames
	const size_t lds = V2LdsSyn::bytes( WAVES ) + ( HOPQ == 0 ? size_t( WAVES ) * size_t( ( p.window_size + 63 ) & ~63 ) * 4 : 0 );   // (HOPQ 0: a ring of `window` floats per wavefront behind the buffers)
	static_assert( V2LdsSyn::bytes( WAVES ) + size_t( WAVES ) * 2048 * 4 <= kMaxLds, "LDS budget" );
	auto kern = k_synthesize_v2<WAVES, HOPQ, ABL>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	static_assert( WAVES == 8, "SynthLayout::groups_per_channel counts groups of 8 chains" );
	const int64_t blocks = int64_t( ( p.chains_per_channel + WAVES - 1 ) / WAVES ) * p.num_channels;
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * WAVES ), lds, s, p, tb );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

template<int TEAMS, int HS, bool WBIG = false>
static int run_synth_eo_team( const SynthParams & p, const FastTables & tb, hipStream_t s )
	{
	using L = typename std::conditional<WBIG, EoLdsBig, EoLds>::type;
	const size_t lds = L::bytes( WBIG ? TEAMS : 2 * TEAMS );                   // two A / B buffer sets per team (windows above 2048: one)
	static_assert( L::bytes( WBIG ? TEAMS : 2 * TEAMS ) <= kMaxLds, "LDS budget" );
	auto kern = k_synthesize_eo_team<TEAMS, HS, WBIG>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + TEAMS - 1 ) / TEAMS ) * p.num_channels;   // a block = a group of TEAMS chains of one channel
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 128 * TEAMS ), lds, s, p, tb );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

// dft 4096, any hop <= window, any window: the team kernels with the overlap-add accumulator as a ring in LDS (pv_kernels_eo.h: HS = -1)
template<int TEAMS, bool WBIG>
static int run_synth_eo_team_ring( const SynthParams & p, const FastTables & tb, hipStream_t s )
	{
	using L = typename std::conditional<WBIG, EoLdsBig, EoLds>::type;
	const size_t lds = L::bytes( TEAMS ) + size_t( WBIG ? TEAMS - 1 : TEAMS ) * size_t( ( p.window_size + 63 ) & ~63 ) * 4;     // one A / B buffer set per team, then the rings (windows above 2048: the last team's in the window table's place)
	FLANHIP_REQUIRE( lds <= kMaxLds && p.window_size <= ( WBIG ? 4096 : 2048 ), FLANHIP_ERR_UNSUPPORTED, "window too long for the LDS ring" );
	auto kern = k_synthesize_eo_team<TEAMS, -1, WBIG>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t blocks = int64_t( ( p.chains_per_channel + TEAMS - 1 ) / TEAMS ) * p.num_channels;   // a block = a group of TEAMS chains of one channel
	FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 128 * TEAMS ), lds, s, p, tb );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

template<int LOG2C>
static int run_synth_fast_hop( const SynthParams & p, const FastTables & tb, hipStream_t s )
	{
	const int kind = synth_fast_kind( 2 << LOG2C, p.window_size, p.hop );
	if constexpr( LOG2C == 10 )
		{
		if( kind == 2 ) return run_synth_v2<8, 0>( p, tb, s );                     // any hop <= window, any window: the accumulator as an LDS ring (round 5)
#ifdef FLANHIP_ABLATIONS
		if( debug_options().syn_variant == 102 && p.hop == 512 ) return run_synth_v2<8, 4, 2>( p, tb, s );
		if( debug_options().syn_variant == 104 && p.hop == 512 ) return run_synth_v2<8, 4, 4>( p, tb, s );
		if( debug_options().syn_variant == 106 && p.hop == 512 ) return run_synth_v2<8, 4, 6>( p, tb, s );
		if( debug_options().syn_variant == 108 && p.hop == 512 ) return run_synth_v2<8, 4, 8>( p, tb, s );
		if( debug_options().syn_variant == 114 && p.hop == 512 ) return run_synth_v2<8, 4, 14>( p, tb, s );
#endif
		switch( p.hop / 128 )
			{
			case 1: return run_synth_v2<8, 1>( p, tb, s );
			case 2: return run_synth_v2<8, 2>( p, tb, s );
			case 4: return run_synth_v2<8, 4>( p, tb, s );
			case 8: return run_synth_v2<8, 8>( p, tb, s );
			}
		return FLANHIP_ERR_UNSUPPORTED;
		}
	else
		{
		// dft 4096: teams of two wavefronts, two 1024-point transforms per frame (pv_kernels_eo.h); windows above 2048 (window = dft is the plain STFT
		// call) the same teams with full-length transforms and one buffer set (WBIG)
		const bool wbig = p.window_size > 2048;
		if( kind == 2 ) return wbig ? run_synth_eo_team_ring<4, true>( p, tb, s ) : run_synth_eo_team_ring<4, false>( p, tb, s );
		switch( p.hop )
			{
			case 128:  return wbig ? run_synth_eo_team<4, 0, true>( p, tb, s ) : run_synth_eo_team<4, 0>( p, tb, s );
			case 256:  return wbig ? run_synth_eo_team<4, 1, true>( p, tb, s ) : run_synth_eo_team<4, 1>( p, tb, s );
			case 512:  return wbig ? run_synth_eo_team<4, 2, true>( p, tb, s ) : run_synth_eo_team<4, 2>( p, tb, s );
			case 1024: return wbig ? run_synth_eo_team<4, 4, true>( p, tb, s ) : run_synth_eo_team<4, 4>( p, tb, s );
			}
		return FLANHIP_ERR_UNSUPPORTED;
		}
	}

static bool synth_fast_ok( int dft, int W, int hop ) { return synth_fast_kind( dft, W, hop ) != 0; }

// Chains per group of the kernels that pass group totals from analysis to synthesis (no scan kernel between the two): 8 for the dft 2048
// pair (a block = 8 one-wavefront chains of a channel), 4 for the dft 4096 team kernels (4 teams per block); 0: no group totals for this shape.
static int self_carry_group( int dft, int W, int hop, int chains_per_channel )
	{
	if( sub_shape( dft, W, hop ) ) return chains_per_channel < 128 ? 0 : sub_group_size( dft );   // (pv_kernels_sub.h: a block's 8 / 16 chains)
	const int kind = synth_fast_kind( dft, W, hop );
	int g = 0;
	// the LDS-ring form of the dft 4096 team synthesis (any hop, any window up to 2048: four teams per block like the analysis) takes the totals too
	if( kind == 2 ) g = dft == 4096 ? ( debug_options().ana11_old ? 0 : 4 ) : group_size_of( dft );
	else if( kind != 1 ) return 0;
	else if( dft == 2048 || v3_size( dft ) ) g = group_size_of( dft );
	else if( dft == 4096 && !debug_options().ana11_old ) g = 4;                   // (windows above 2048: the WBIG variants)
	// any number of groups (their carries come from a scan of their own); with few chains per channel the scan over the chains themselves is as
	// short and the groups' epilogue and prologue are pure cost (a 5 s mono file: 118 chains, 68 against 73 us per round trip)
	if( chains_per_channel < 128 ) g = 0;
	return g;
	}

// flanhip_analyze_dev_fused always leaves convert_to_audio's pre-pass in the workspace.  The tuned kernels and the generic
// ones up to dft 2048 accumulate the sums while they have every f in a register; the generic kernels for dft >= 4096 keep no
// such state (64 bins per lane), so there the pre-pass kernel itself is run on the fresh PV on the analysis' behalf.
static int launch_analyze_body( const float * d_audio, int64_t ch, int64_t n, float sr, int W, int hop, int dft, flanhip_MF * d_out, void * d_fused_ws, hipStream_t s, bool * left_group_sums, int * epoch );

// The note "this workspace holds group totals too" (which lets the dft 2048 synthesis skip its scan kernel) is written only once the producing
// launch has been accepted, and withdrawn before it is attempted: a launch that fails leaves the workspace marked as needing the scan.
int launch_analyze( const float * d_audio, int64_t ch, int64_t n, float sr, int W, int hop, int dft, flanhip_MF * d_out, void * d_fused_ws, hipStream_t s )
	{
	bool left_group_sums = false;
	int epoch = 0;
	if( d_fused_ws ) note_workspace_producer( d_fused_ws, 0 );
	const int rc = launch_analyze_body( d_audio, ch, n, sr, W, hop, dft, d_out, d_fused_ws, s, &left_group_sums, &epoch );
	if( !rc && d_fused_ws && left_group_sums ) note_workspace_producer( d_fused_ws, 1, epoch );
	return rc;
	}

static int launch_analyze_body( const float * d_audio, int64_t ch, int64_t n, float sr, int W, int hop, int dft, flanhip_MF * d_out, void * d_fused_ws, hipStream_t s, bool * left_group_sums, int * epoch )
	{
	FLANHIP_REQUIRE( d_audio && d_out, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	FLANHIP_REQUIRE( ch > 0 && n >= 0 && W >= 2 && hop >= 1 && sr > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad sizes" );
	FLANHIP_REQUIRE( W <= dft, FLANHIP_ERR_INVALID_ARG, "window_size larger than dft_size" );
	FLANHIP_REQUIRE( dft_size_ok( dft ), FLANHIP_ERR_UNSUPPORTED, "dft_size must be even, at least 4 and at most 2^20" );
	if( int rc = require_device() ) return rc;
	// dft 8192 / 16384 on the team kernels' grid of windows and hops (pv_kernels_team.h, round 6): before the round-1 / mixed-radix kernels of those sizes
	const bool team = team_shape( dft, W, hop ) && n >= 2 && n < ( int64_t( 1 ) << 31 ) - 65536;
	// dft 512 / 256 on the grid of the kernels with several chains per wavefront (pv_kernels_sub.h, round 6): before the one-wavefront / generic kernels
	const bool sub = sub_shape( dft, W, hop ) && n >= 2 && n < ( int64_t( 1 ) << 31 ) - 8192;
	MrPlan mr_plan{};
	const bool mr = !team && mr_size( dft, W, &mr_plan );
	BsPlan bs_plan{};
	const bool bs = !team && !mr && bs_size( dft, W, &bs_plan );
	BigPlan big_plan{};
	const bool big = !team && !mr && !bs && big_size( dft, W, &big_plan );
	const bool any = !team && !fft_size( dft ) && !mr && !bs && !big;
	std::shared_ptr<const PlanRef> plan_ref;                                        // (held until the kernels below are launched)
	if( int rc = get_plan( W, dft, &plan_ref ) ) return rc;
	const Plan * plan = &plan_ref->plan;

	AnalyzeParams p{};
	p.audio = d_audio; p.out = reinterpret_cast<MF*>( d_out );
	p.window = plan->d_window; p.tw = plan->d_tw; p.tw2 = plan->d_tw2; p.dump = plan->d_dump;
	p.n = n; p.F = n / hop + 1;                                   // AudioPV.cpp:17
	p.num_channels = int( ch ); p.window_size = W; p.hop = hop;
	// (hop: the dft 2048 kernel addresses a block's samples by 32-bit byte offsets from the block's first frame -- up to 8 chains of <= ~512 frames,
	// env overrides aside -- and the generic kernels serve the hops that would not fit: nothing anybody analyses with)
	const bool fast = ( dft == 2048 || ( dft == 4096 && !debug_options().ana11_old ) || v3_size( dft ) ) && n >= 2 && n < ( int64_t( 1 ) << 31 ) - 8192 && hop <= 65536 && !force_generic();
	int target_chains = team ? team_target_chains( dft ) : sub ? sub_target_chains( dft ) : any ? any_target_chains( dft / 2 + 1 ) : mr ? mr_target_chains( dft, W ) : bs ? bs_target_chains( dft, W ) : big ? big_target_chains() : fast ? fast_target_chains( dft, false ) : generic_target_chains( dft );
	const int block_group = team ? team_group_size( dft ) : sub ? sub_group_size( dft ) : fast ? group_size_of( dft ) : 1;      // chains per block of the kernels this call runs
	p.L = choose_chain_length( ch, p.F, any ? 7 : 1, target_chains, block_group );
	p.chains_per_channel = int( ( p.F + p.L - 1 ) / p.L );
	p.sample_rate = sr;
	p.analysis_rate = sr / hop;                                   // AudioPV.cpp:26 (float / int)
	DivPlan dp;
	if( int rc = get_div_plan( p.analysis_rate, &dp ) ) return rc;
	p.ar_div = DivC{ dp.c, dp.rc, dp.exact };
	p.sums = nullptr; p.nan_out = nullptr; p.nan_epoch = 0;
	p.cancel = thread_cancel_word( s );                                           // kernels stop starting chains when the thread's wait raises it (core.hip)
	// windows above 2048: the WBIG team kernels (pv_kernels_eo.h)
	bool kernel_sums = !any && !big && ( !mr || mr_pingpong( mr_plan.C ) ) && ( !bs || bs_pingpong( bs_plan.M ) );   // every power-of-two analysis kernel keeps the sums, and the mixed-radix one where its LDS has room; for the rest the pre-pass kernel runs on the analysis' behalf
	SynthLayout fused_lay{};
	if( d_fused_ws )
		{
		// use the chain layout convert_to_audio will use for this PV and leave its pre-pass results in the workspace
		SynthLayout & lay = fused_lay;
		if( int rc = synth_layout( ch, p.F, dft / 2 + 1, sr, p.analysis_rate, W, &lay ) ) return rc;
		p.nan_epoch = next_epoch();
		*epoch = p.nan_epoch;
		// A SHORT input at a large window / hop ratio (the reference API's default call on a few seconds of sound: ( 2048, 128, 4096 ), ratio 16): the synthesis' chains
		// are at least window / hop - 1 = 15 frames long, the analysis' own cut would be 4 -- walking the synthesis' chains costs the launch eleven frames of latency
		// (2 ch x 5 s: 0.196 ms fused against 0.150 as two plain calls, tools/unfused_sweep.py).  There the analysis keeps its own cut and the pre-pass kernel sums
		// the rows on its behalf, like behind the kernels that keep no sums.
		if( kernel_sums && lay.L >= p.L + 8 && int64_t( p.chains_per_channel ) * ch <= target_chains ) kernel_sums = false;
		if( kernel_sums )
			{
			p.L = lay.L;
			p.chains_per_channel = lay.chains_per_channel;
			p.sums = reinterpret_cast<double*>( d_fused_ws );
			p.nan_out = reinterpret_cast<int*>( reinterpret_cast<char*>( d_fused_ws ) + lay.carry_bytes + lay.head_bytes );
			}
		// the dft 2048 kernel (and the dft 4096 team kernel) also leaves one total per group of 8 (4) chains: the synthesis kernel then needs no scan kernel in front of it
		const bool groups_too = ( fast || sub ) && kernel_sums && self_carry_group( dft, W, hop, lay.chains_per_channel ) != 0;
		p.group_sums = groups_too ? reinterpret_cast<double*>( reinterpret_cast<char*>( d_fused_ws ) + lay.group_offset ) : nullptr;
		p.groups_per_channel = lay.groups_per_channel;
		*left_group_sums = groups_too;
		}
	auto prepass_on_behalf = [&]() -> int                                         // for an analysis kernel that keeps no sums (none does at present)
		{
		if( !d_fused_ws || kernel_sums ) return FLANHIP_OK;
		SynthParams q{};
		q.pv = reinterpret_cast<const MF*>( d_out );
		q.carry = reinterpret_cast<double*>( d_fused_ws );
		q.F = p.F; q.num_channels = int( ch ); q.num_bins = dft / 2 + 1;
		q.L = fused_lay.L; q.chains_per_channel = fused_lay.chains_per_channel;
		q.analysis_rate = p.analysis_rate; q.ar_div = p.ar_div;
		q.nan_words = reinterpret_cast<int*>( reinterpret_cast<char*>( d_fused_ws ) + fused_lay.carry_bytes + fused_lay.head_bytes );
		q.nan_epoch = p.nan_epoch;
		const int64_t chains = int64_t( q.chains_per_channel ) * ch;
		hipLaunchKernelGGL( k_phase_sums2, dim3( (unsigned) chains, (unsigned) ( ( q.num_bins + 255 ) / 256 ) ), dim3( 256 ), 0, s, q );
		FLANHIP_CHECK( hipGetLastError() );
		return FLANHIP_OK;
		};

	if( team ) { if( int rc = run_analyze_team( p, *plan, dft, s ) ) return rc; return prepass_on_behalf(); }    // (keeps the chain sums where kernel_sums says so)
	if( sub ) { if( int rc = run_analyze_sub( p, dft, s ) ) return rc; return prepass_on_behalf(); }
	if( any )
		{
		std::shared_ptr<const UnitRef> unit_ref;
		if( int rc = get_unit_circle( dft, &unit_ref ) ) return rc;
		const d2 * unit = unit_ref->d;
		const int64_t chains = int64_t( p.chains_per_channel ) * ch;
		const int bin_blocks = ( dft / 2 + 1 + ANY_THREADS - 1 ) / ANY_THREADS;
		FLANHIP_REQUIRE( chains <= 65535, FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
		hipLaunchKernelGGL( k_analyze_any, dim3( (unsigned) bin_blocks, (unsigned) chains ), dim3( ANY_THREADS ), 0, s, p, unit, dft );
		FLANHIP_CHECK( hipGetLastError() );
		return prepass_on_behalf();
		}
	if( mr )
		{
		const int64_t chains = int64_t( p.chains_per_channel ) * ch;
		FLANHIP_REQUIRE( chains < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
		const size_t lds = mr_analyze_lds( mr_plan.C, W, mr_plan.win_lds, mr_plan.kc_lds );
		FLANHIP_REQUIRE( mr_plan_fits_kernels( mr_plan ), FLANHIP_ERR_UNSUPPORTED, "mixed-radix plan and kernels disagree" );
		auto kern = !mr_pingpong( mr_plan.C ) ? k_analyze_mr<false, true> : mr_plan_is_big( mr_plan ) ? k_analyze_mr<true, true> : k_analyze_mr<true, false>;
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( kern, dim3( (unsigned) chains ), dim3( MR_THREADS ), lds, s, p, mr_plan );
		FLANHIP_CHECK( hipGetLastError() );
		return prepass_on_behalf();
		}
	if( big )
		{
		const int64_t blocks = big_blocks( int64_t( p.chains_per_channel ) * ch, big_plan.P );
		FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
		const size_t lds = big_analyze_lds( big_plan.C2 );
		const int bq = ( big_plan.C2 + MR_THREADS - 1 ) / MR_THREADS;                 // bins of a residue per thread
		auto kern = big_plan.mixed == 2 ? ( bq > 4 ? k_analyze_big<8, 2> : bq > 2 ? k_analyze_big<4, 2> : k_analyze_big<2, 2> )
			: big_plan.mixed ? ( bq > 4 ? k_analyze_big<8, 1> : bq > 2 ? k_analyze_big<4, 1> : k_analyze_big<2, 1> )
			: big_plan.C2 == 4096 ? k_analyze_big<8> : big_plan.C2 == 2048 ? k_analyze_big<4> : k_analyze_big<2>;
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( MR_THREADS ), lds, s, p, big_plan );
		FLANHIP_CHECK( hipGetLastError() );
		return prepass_on_behalf();
		}
	if( bs )
		{
		const int64_t chains = int64_t( p.chains_per_channel ) * ch;
		FLANHIP_REQUIRE( chains < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
		FLANHIP_REQUIRE( plan->d_bs_tw && plan->d_bs_chirp && plan->d_bs_bh, FLANHIP_ERR_UNSUPPORTED, "chirp-z plan and tables disagree" );
		BsTables g{ plan->d_bs_tw, plan->d_bs_chirp, plan->d_bs_bh };
		if( bs_plan.glob )
			{
			// buffers and state of every block in device memory: convert_to_PV has no workspace argument, so the stretch is the stream's own (allocated and freed in
			// stream order around the launch)
			g.scratch_stride = bsg_block_bytes( bs_plan.C, bs_plan.M, W );
			void * scratch = nullptr;
			FLANHIP_CHECK( hipMallocAsync( &scratch, g.scratch_stride * size_t( chains ), s ) );
			g.scratch = static_cast<unsigned char*>( scratch );
			hipLaunchKernelGGL( ( k_analyze_bs<true, false, true> ), dim3( (unsigned) chains ), dim3( MR_THREADS ), 0, s, p, bs_plan, g );
			const hipError_t launched = hipGetLastError();
			FLANHIP_CHECK( hipFreeAsync( scratch, s ) );
			FLANHIP_CHECK( launched );
			return prepass_on_behalf();
			}
		const size_t lds = bs_analyze_lds( bs_plan.C, bs_plan.M );
		auto kern = !bs_pingpong( bs_plan.M ) ? k_analyze_bs<false, false> : bs_hoist( bs_plan, W ) ? k_analyze_bs<true, true> : k_analyze_bs<true, false>;
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( kern, dim3( (unsigned) chains ), dim3( MR_THREADS ), lds, s, p, bs_plan, g );
		FLANHIP_CHECK( hipGetLastError() );
		return prepass_on_behalf();
		}
	if( fast )
		{
		FastTables tb{ plan->d_tw1f, plan->d_tw3f, plan->d_tw2 };
		int rc;
		if( v3_size( dft ) ) rc = p.sums ? run_analyze_v3_cfg<true>( dft, p, s ) : run_analyze_v3_cfg<false>( dft, p, s );
		else if( dft == 2048 ) rc = p.sums ? run_analyze_v2_variant<true>( debug_options().ana_variant, p, tb, s ) : run_analyze_v2_variant<false>( debug_options().ana_variant, p, tb, s );
		// teams of two wavefronts, two E / O buffer sets, one meeting per frame (0.27 ms for 8 ch x 60 s with the fused round trip's chain sums,
		// 0.28 without; one set and two meetings: +7 %; the round-1 kernel: 0.44)
		else if( W <= 2048 ) rc = p.sums ? run_analyze_eo_team<4, true, 2, true>( p, tb, s ) : run_analyze_eo_team<4, false, 2, true>( p, tb, s );
		// windows up to the whole transform: the same decomposition with full-length E / O inputs, one buffer set (pv_kernels_eo.h: WBIG)
		else rc = p.sums ? run_analyze_eo_team<4, true, 2, false, true>( p, tb, s ) : run_analyze_eo_team<4, false, 2, false, true>( p, tb, s );
		if( rc ) return rc;
		return prepass_on_behalf();                                                 // (a short input on its own cut: the sums by the pre-pass kernel)
		}

	int rc = FLANHIP_ERR_UNSUPPORTED;
	switch( ilog2( dft ) - 1 )
		{
		case 4:  rc = run_analyze<4, 4>( p, s ); break;
		case 5:  rc = run_analyze<5, 4>( p, s ); break;
		case 6:  rc = run_analyze<6, 4>( p, s ); break;
		case 7:  rc = run_analyze<7, 4>( p, s ); break;
		case 8:  rc = run_analyze<8, 4>( p, s ); break;
		case 9:  rc = run_analyze<9, 1, 2>( p, s ); break;               // teams of 2 / 4 / 4 / 8 wavefronts per chain: measured best
		case 10: rc = run_analyze<10, 1, 4>( p, s ); break;
		case 11: rc = run_analyze<11, 1, 4>( p, s ); break;
		case 12: rc = run_analyze<12, 1, kTeamWaves12>( p, s ); break;   // a block of 8 wavefronts per chain
		default: set_error( "unsupported dft_size %d", dft ); break;
		}
	if( rc ) return rc;
	return prepass_on_behalf();
	}

int synth_layout( int64_t ch, int64_t F, int bins, float sr, float ar, int W, SynthLayout * o )
	{
	FLANHIP_REQUIRE( ch > 0 && F > 0 && bins >= 2 && W >= 2 && sr > 0.0f && ar > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad sizes" );
	o->dft = ( bins - 1 ) * 2;                                    // PVBuffer.cpp:356-359
	o->hop = int( sr / ar );                                      // PVBuffer.cpp:381-384
	FLANHIP_REQUIRE( o->hop >= 1, FLANHIP_ERR_INVALID_ARG, "analysis_rate above sample_rate: hop size 0" );
	FLANHIP_REQUIRE( W <= o->dft, FLANHIP_ERR_INVALID_ARG, "window_size larger than dft size" );
	FLANHIP_REQUIRE( dft_size_ok( o->dft ), FLANHIP_ERR_UNSUPPORTED, "dft size must be even, at least 4 and at most 2^20" );
	FLANHIP_REQUIRE( int64_t( o->dft ) * W < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "dft_size * window_size overflows the int product of AudioPV.cpp:99" );
	const bool team = team_shape( o->dft, W, o->hop );
	const bool sub = sub_shape( o->dft, W, o->hop );
	const bool mr = !team && mr_size( o->dft, W );
	const bool bs = !team && !mr && bs_size( o->dft, W );
	BigPlan big_plan{};
	o->big = !team && !mr && !bs && big_size( o->dft, W, &big_plan );
	o->any = !team && !fft_size( o->dft ) && !mr && !bs && !o->big;
	o->head_len = o->any ? 0 : std::max( W - o->hop, 0 );        // (the direct-sum path overlap-adds whole frames from its own scratch: no chain heads)
	const int overlap = ( W + o->hop - 1 ) / o->hop;              // frames covering one output sample
	const int kind = ( o->any || o->big || sub ) ? 0 : synth_fast_kind( o->dft, W, o->hop );
	const int slots = team ? team_target_chains( o->dft ) : sub ? sub_target_chains( o->dft ) : o->any ? any_target_chains( bins ) : mr ? mr_target_chains( o->dft, W ) : bs ? bs_target_chains( o->dft, W ) : o->big ? big_target_chains() : kind == 0 ? generic_target_chains( o->dft ) : fast_target_chains( o->dft, true );
	const int block_group = team ? team_group_size( o->dft ) : sub ? sub_group_size( o->dft ) : kind != 0 ? group_size_of( o->dft ) : 1;
	o->L = choose_chain_length( ch, F, o->any ? 1 : std::max( overlap - 1, 1 ), slots, block_group );
	o->chains_per_channel = int( ( F + o->L - 1 ) / o->L );
	const int64_t chains = int64_t( o->chains_per_channel ) * ch;
	o->carry_bytes = ( size_t( chains ) * bins * sizeof( double ) + 255 ) & ~size_t( 255 );
	o->head_bytes = ( size_t( chains ) * o->head_len * sizeof( float ) + 255 ) & ~size_t( 255 );
	const int gsize = o->any ? 8 : sub ? sub_group_size( o->dft ) : group_size_of( o->dft );        // self_carry_group
	o->groups_per_channel = ( o->chains_per_channel + gsize - 1 ) / gsize;
	o->group_offset = o->carry_bytes + o->head_bytes + 1024;       // tail: NaN flag (4 B at +0), dump area (512 B at +512); then the group sums
	o->group_bytes = ( size_t( ch ) * o->groups_per_channel * bins * sizeof( double ) + 255 ) & ~size_t( 255 );
	o->total_bytes = o->group_offset + 2 * o->group_bytes;             // the producer's group totals, then the group carries the synthesis' own scan over them leaves
	// the dft 2048 / 1024 / 512 synthesis kernels add the overlaps of neighbouring chains themselves (pv_kernels_v2.h, _v3.h): a state word per chain and a second side buffer
	o->fix_offset = o->tail_offset = 0;
	// (round 6: the dft 4096 team synthesis too, a word per WAVEFRONT of a chain: pv_kernels_eo.h)
	if( ( team || ( !o->any && !mr && ( o->dft == 2048 || v3_size( o->dft ) || o->dft == 4096 ) && kind == 1 ) ) && o->head_len > 0 )
		{
		o->fix_offset = o->total_bytes;
		o->tail_offset = o->fix_offset + ( ( size_t( chains ) * 8 * sizeof( int ) + 255 ) & ~size_t( 255 ) );
		o->total_bytes = o->tail_offset + o->head_bytes;
		}
	o->any_spec_offset = o->any_frames_offset = 0;
	if( o->any )
		{
		// scratch of the direct-sum synthesis: the spectra X[ch][F][bins] (one PV's worth) and the windowed frames [ch][F][W]
		o->any_spec_offset = o->total_bytes;
		o->any_frames_offset = o->any_spec_offset + ( ( size_t( ch ) * size_t( F ) * size_t( bins ) * sizeof( cf ) + 255 ) & ~size_t( 255 ) );
		o->total_bytes = o->any_frames_offset + ( ( size_t( ch ) * size_t( F ) * size_t( W ) * sizeof( float ) + 255 ) & ~size_t( 255 ) );
		}
	o->bsg_offset = 0;
	if( BsPlan bp{}; bs && bs_size( o->dft, W, &bp ) && bp.glob )
		{
		// the chirp-z kernels' buffers and state in device memory (BsPlan::glob: M above 8192), a stretch per chain
		o->bsg_offset = o->total_bytes;
		o->total_bytes += bsg_block_bytes( bp.C, bp.M, W ) * size_t( chains );
		}
	o->mr_ring_offset = 0;
	if( MrPlan mp{}; mr && mr_size( o->dft, W, &mp ) && mp.ring_ws )
		{
		// the mixed-radix synthesis' overlap-add rings [chains][W], where they do not fit the LDS (MrPlan::ring_ws)
		o->mr_ring_offset = o->total_bytes;
		o->total_bytes += ( size_t( chains ) * size_t( ( W + 3 ) & ~3 ) * sizeof( float ) + 255 ) & ~size_t( 255 );
		}
	o->big_out_offset = o->big_head_offset = o->big_ring_offset = 0;
	if( o->big )
		{
		// the units' partial output streams [P][ch][F hop] and heads [P][chains][head_len] (pv_kernels_big.h)
		o->big_out_offset = o->total_bytes;
		o->big_head_offset = o->big_out_offset + ( ( size_t( big_plan.P ) * size_t( ch ) * size_t( F ) * size_t( o->hop ) * sizeof( float ) + 255 ) & ~size_t( 255 ) );
		o->total_bytes = o->big_head_offset + ( ( size_t( big_plan.P ) * size_t( chains ) * size_t( o->head_len ) * sizeof( float ) + 255 ) & ~size_t( 255 ) );
		if( big_ring_in_workspace( big_plan, W ) )
			{
			o->big_ring_offset = o->total_bytes;
			o->total_bytes += ( size_t( big_plan.P ) * size_t( chains ) * size_t( ( W + 3 ) & ~3 ) * sizeof( float ) + 255 ) & ~size_t( 255 );
			}
		}
	o->flags_offset = o->total_bytes;
	o->total_bytes += ( sizeof( int ) * size_t( bins + 1 ) + 255 ) & ~size_t( 255 );
	return FLANHIP_OK;
	}

template<int LOG2C, int WAVES, int T = 1>
static int run_synth( const SynthParams & p, hipStream_t s )
	{
	const size_t lds = synth_lds_bytes( 1 << LOG2C, p.window_size, WAVES, LOG2C >= 12 && T == 1 );
	FLANHIP_REQUIRE( lds <= kMaxLds, FLANHIP_ERR_UNSUPPORTED, "window/dft too large for LDS" );
	auto kern = k_synthesize<LOG2C, WAVES, T>;
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	const int64_t chains = int64_t( p.chains_per_channel ) * p.num_channels;
	const int64_t blocks = ( chains + WAVES - 1 ) / WAVES;
	hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( 64 * WAVES * T ), lds, s, p );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int launch_synthesize( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W, float * d_out,
	void * d_ws, int * d_nan, int presummed, hipStream_t s, const double * d_carry_in, double * d_total_out, bool prepass_only, int stage_mask )
	{
	FLANHIP_REQUIRE( d_pv && ( d_out || prepass_only ) && d_ws, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	SynthLayout lay;
	if( int rc = synth_layout( ch, F, bins, sr, ar, W, &lay ) ) return rc;
	if( int rc = require_device() ) return rc;
	std::shared_ptr<const PlanRef> plan_ref;                                        // (held until the kernels below are launched)
	if( int rc = get_plan( W, lay.dft, &plan_ref ) ) return rc;
	const Plan * plan = &plan_ref->plan;

	SynthParams p{};                                               // every optional pointer null unless set below
	p.pv = reinterpret_cast<const MF*>( d_pv ); p.out = d_out;
	p.window = plan->d_window; p.tw = plan->d_tw; p.tw2 = plan->d_tw2;
	p.carry = reinterpret_cast<double*>( d_ws );
	p.head = reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.carry_bytes );
	p.nan_flag = d_nan;
	p.F = F; p.out_len = F * lay.hop;                              // AudioPV.cpp:93
	p.num_channels = int( ch ); p.window_size = W; p.hop = lay.hop; p.L = lay.L;
	p.chains_per_channel = lay.chains_per_channel; p.head_len = lay.head_len; p.num_bins = bins;
	p.analysis_rate = ar;
	p.window_scale = 2.67f / ( lay.dft * W / lay.hop );            // AudioPV.cpp:99 (integer arithmetic in the divisor)
	DivPlan dp;
	if( int rc = get_div_plan( ar, &dp ) ) return rc;
	p.ar_div = DivC{ dp.c, dp.rc, dp.exact };
	p.dump = reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.carry_bytes + lay.head_bytes + 512 );
	p.nan_in = ( presummed == 1 || presummed == 2 ) ? reinterpret_cast<const int*>( reinterpret_cast<char*>( d_ws ) + lay.carry_bytes + lay.head_bytes ) : nullptr;
	p.skip_words = presummed == 2 ? p.nan_in : nullptr;
	p.carry_in = d_carry_in; p.total_out = d_total_out; p.total_only = prepass_only ? 1 : 0;
	p.cancel = thread_cancel_word( s );
	// the register-accumulator synthesis kernels (dft 512 ... 16384 on their grids) add the chains' overlaps themselves where their chains are long enough to publish their heads from inside the frame
	// loop (round 5: -2 % of the bench shape's step, -6 % of the stereo minute's; chains of a few frames would publish at their ends and pay
	// the exchange there: +8 % at 1 ch x 5 s, so those keep k_ola_fixup as a launch of its own).  FLANHIP_DEBUG_INLINE_FIXUP: 1 always, 2 never.
	const int fix_hook = debug_options().inline_fixup;
	const bool self_fix = lay.fix_offset != 0 && !prepass_only && ( stage_mask & 0xF ) == 0xF && fix_hook != 2
		&& ( fix_hook == 1 || lay.L >= ( lay.head_len + lay.hop - 1 ) / lay.hop + 3 );
	if( self_fix )
		{
		p.fix_state = reinterpret_cast<int*>( reinterpret_cast<char*>( d_ws ) + lay.fix_offset );
		p.tail = reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.tail_offset );
		p.fix_tag = int( ( unsigned( next_epoch() ) & 0x1FFFFFFFu ) << 2 );
		}

	const int64_t chains = int64_t( p.chains_per_channel ) * ch;
	FLANHIP_REQUIRE( chains < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
	const int stages = prepass_only ? 3 : ( stage_mask & 0xF );
	// The dft 2048 analysis kernel leaves group totals beside the chain sums (launch_analyze notes that for this workspace): the dft 2048
	// synthesis kernel then works out its own carries and the scan kernel is not launched.  Any other producer or shape: the scan runs.
	int noted_epoch = 0;
	const bool self_carry = presummed == 1 && !prepass_only && !d_carry_in && !d_total_out && debug_options().syn_variant != 2
		&& self_carry_group( lay.dft, W, lay.hop, lay.chains_per_channel ) != 0 && workspace_producer( d_ws, &noted_epoch ) == 1;
	if( self_carry )
		{
		// the note is the host's; what is IN the workspace is checked by the kernel against it (two callers racing on one workspace: flag 2, not silence)
		p.expect_epoch = noted_epoch;
		p.group_sums = reinterpret_cast<const double*>( reinterpret_cast<char*>( d_ws ) + lay.group_offset );
		p.groups_per_channel = lay.groups_per_channel;
		// up to 40 groups per channel the synthesis kernel adds the totals of the groups before its own itself (no kernel in front: measured
		// cheaper, the scan kernel's floor is ~9 us); beyond, a scan over the group totals leaves every group's carry
		const bool scan_groups = p.groups_per_channel > 40;
		p.group_carry = scan_groups ? reinterpret_cast<double*>( reinterpret_cast<char*>( d_ws ) + lay.group_offset + lay.group_bytes ) : nullptr;
		if( scan_groups && ( stages & 2 ) )
			{
			// the scan over the GROUP totals (an eighth / a quarter of the chains): every group's carry, from which the synthesis kernel's prologue
			// and the chain sums give every chain's
			if( p.groups_per_channel <= 512 ) hipLaunchKernelGGL( ( k_phase_scan2<16, true> ), dim3( (unsigned) ( ( bins + 31 ) / 32 ), (unsigned) ch ), dim3( 512 ), 0, s, p );
			else hipLaunchKernelGGL( ( k_phase_scan2<32, true> ), dim3( (unsigned) ( ( bins + 15 ) / 16 ), (unsigned) ch ), dim3( 512 ), 0, s, p );
			FLANHIP_CHECK( hipGetLastError() );
			}
		}
	// Chain sums but no group totals (the pre-pass, or a producer that keeps the sums only: PV::modify_time, PV::shape): a small kernel adds
	// up the groups, and the synthesis kernel works out its carries as above -- instead of the scan over all the chains (config 3: 5 us for 16)
	const int gsize = self_carry_group( lay.dft, W, lay.hop, lay.chains_per_channel );
	const bool group_kernel = !self_carry && gsize != 0 && stages == 0xF && !prepass_only && !d_carry_in && !d_total_out && debug_options().syn_variant != 2 && presummed != 3;
	// ... and where the chain sums are still to be made (or may have been handed over: presummed 2), the pre-pass and that kernel are ONE launch
	const bool sums_in_group_kernel = group_kernel && presummed != 1;
	if( ( stages & 1 ) && presummed != 1 && presummed != 3 && !sums_in_group_kernel )
		{
		hipLaunchKernelGGL( k_phase_sums2, dim3( (unsigned) chains, (unsigned) ( ( bins + 255 ) / 256 ) ), dim3( 256 ), 0, s, p );
		FLANHIP_CHECK( hipGetLastError() );
		}
	if( group_kernel )
		{
		double * gs = reinterpret_cast<double*>( reinterpret_cast<char*>( d_ws ) + lay.group_offset );
		p.groups_per_channel = lay.groups_per_channel;
		const dim3 grid( (unsigned) ( ( bins + 255 ) / 256 ), (unsigned) lay.groups_per_channel, (unsigned) ch );
		FLANHIP_REQUIRE( lay.groups_per_channel <= 65535 && ch <= 65535, FLANHIP_ERR_UNSUPPORTED, "too many groups / channels for one launch" );
		if( sums_in_group_kernel ) { if( gsize == 32 ) hipLaunchKernelGGL( k_sums_and_groups<32>, grid, dim3( 256 ), 0, s, p, gs ); else if( gsize == 16 ) hipLaunchKernelGGL( k_sums_and_groups<16>, grid, dim3( 256 ), 0, s, p, gs ); else if( gsize == 8 ) hipLaunchKernelGGL( k_sums_and_groups<8>, grid, dim3( 256 ), 0, s, p, gs ); else hipLaunchKernelGGL( k_sums_and_groups<4>, grid, dim3( 256 ), 0, s, p, gs ); }
		else if( gsize == 32 ) hipLaunchKernelGGL( k_group_sums<32>, grid, dim3( 256 ), 0, s, p, gs ); else if( gsize == 16 ) hipLaunchKernelGGL( k_group_sums<16>, grid, dim3( 256 ), 0, s, p, gs ); else if( gsize == 8 ) hipLaunchKernelGGL( k_group_sums<8>, grid, dim3( 256 ), 0, s, p, gs ); else hipLaunchKernelGGL( k_group_sums<4>, grid, dim3( 256 ), 0, s, p, gs );
		FLANHIP_CHECK( hipGetLastError() );
		p.group_sums = gs;
		const bool scan_groups = p.groups_per_channel > 40;
		p.group_carry = scan_groups ? reinterpret_cast<double*>( reinterpret_cast<char*>( d_ws ) + lay.group_offset + lay.group_bytes ) : nullptr;
		if( scan_groups )
			{
			if( p.groups_per_channel <= 512 ) hipLaunchKernelGGL( ( k_phase_scan2<16, true> ), dim3( (unsigned) ( ( bins + 31 ) / 32 ), (unsigned) ch ), dim3( 512 ), 0, s, p );
			else hipLaunchKernelGGL( ( k_phase_scan2<32, true> ), dim3( (unsigned) ( ( bins + 15 ) / 16 ), (unsigned) ch ), dim3( 512 ), 0, s, p );
			FLANHIP_CHECK( hipGetLastError() );
			}
		}
	if( ( stages & 2 ) && !self_carry && !group_kernel )
		{
		const int64_t cols = ch * bins;
		// segments of at most 32 chains where that keeps 16 bins (128 contiguous bytes) per block row (pv_kernels_fast.h; 8 bins per block: slower)
		// few chains per channel and rows of thousands of bins (the dft 8192 / 16384 team kernels' layouts): a thread per column (round 6)
		// ... or many channels of few chains each (a batch of short clips: 1024 channels x 8 chains took k_phase_scan2 0.39 ms -- 33 792 blocks of 512 threads with
		// half a segment's work each -- and this one 0.0x: tools/short_channel_experiment.py)
		const bool flat = p.chains_per_channel <= 64 && ( bins >= 2049 || cols >= 65536 ) && !d_carry_in && !d_total_out && !prepass_only;
		if( flat && p.chains_per_channel <= 32 ) hipLaunchKernelGGL( k_phase_scan_flat<32>, dim3( (unsigned) ( ( bins + 255 ) / 256 ), (unsigned) ch ), dim3( 256 ), 0, s, p );
		else if( flat ) hipLaunchKernelGGL( k_phase_scan_flat<64>, dim3( (unsigned) ( ( bins + 255 ) / 256 ), (unsigned) ch ), dim3( 256 ), 0, s, p );
		else if( p.chains_per_channel <= 512 ) hipLaunchKernelGGL( k_phase_scan2<16>, dim3( (unsigned) ( ( bins + 31 ) / 32 ), (unsigned) ch ), dim3( 512 ), 0, s, p );
		else if( p.chains_per_channel <= 1024 ) hipLaunchKernelGGL( k_phase_scan2<32>, dim3( (unsigned) ( ( bins + 15 ) / 16 ), (unsigned) ch ), dim3( 512 ), 0, s, p );
		else hipLaunchKernelGGL( k_phase_scan2<64>, dim3( (unsigned) ( ( bins + 7 ) / 8 ), (unsigned) ch ), dim3( 512 ), 0, s, p );
		FLANHIP_CHECK( hipGetLastError() );
		}

	int rc = FLANHIP_ERR_UNSUPPORTED;
	if( !( stages & 4 ) ) rc = FLANHIP_OK;
	else if( lay.any )
		{
		// any even dft size without FFT kernels: spectra along the chains, c2r by its definition, overlap-add in frame order (pv_kernels_any.h)
		std::shared_ptr<const UnitRef> unit_ref;
		if( int rc2 = get_unit_circle( lay.dft, &unit_ref ) ) return rc2;
		const d2 * unit = unit_ref->d;
		AnySynthParams q{};
		q.pv = p.pv; q.carry = p.carry; q.out = d_out; q.window = p.window;
		q.spec = reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.any_spec_offset );
		q.frames = reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.any_frames_offset );
		q.F = F; q.out_len = p.out_len; q.num_channels = int( ch ); q.bins = bins; q.N = lay.dft; q.W = W; q.hop = lay.hop;
		q.L = lay.L; q.chains_per_channel = lay.chains_per_channel; q.analysis_rate = ar; q.window_scale = p.window_scale;
		q.cancel = p.cancel;
		const unsigned bin_blocks = (unsigned) ( ( bins + ANY_THREADS - 1 ) / ANY_THREADS );
		FLANHIP_REQUIRE( chains <= 65535, FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
		const int64_t batches = ( ch * F + ANY_FB - 1 ) / ANY_FB;
		FLANHIP_REQUIRE( batches <= 65535 * int64_t( 32768 ), FLANHIP_ERR_UNSUPPORTED, "too many frames for one launch" );
		hipLaunchKernelGGL( k_any_spectra, dim3( bin_blocks, (unsigned) chains ), dim3( ANY_THREADS ), 0, s, q );
		FLANHIP_CHECK( hipGetLastError() );
		// frame batches over grid.y (<= 65535 per launch): long PVs in several launches of the same kernel over consecutive row ranges
		const unsigned sample_blocks = (unsigned) ( ( W + ANY_THREADS - 1 ) / ANY_THREADS );
		for( int64_t b0 = 0; b0 < batches; b0 += 65535 )
			{
			AnySynthParams qq = q;
			const int64_t rows0 = b0 * ANY_FB, nb = std::min<int64_t>( 65535, batches - b0 );
			qq.spec = q.spec + size_t( rows0 ) * size_t( bins ) * 2;
			qq.frames = q.frames + size_t( rows0 ) * size_t( W );
			// ( num_channels * F is only used as the row count there: hand the rows this launch owns )
			qq.num_channels = 1; qq.F = std::min<int64_t>( ch * F - rows0, nb * ANY_FB );
			hipLaunchKernelGGL( k_any_inverse, dim3( sample_blocks, (unsigned) nb ), dim3( ANY_THREADS ), 0, s, qq, unit );
			FLANHIP_CHECK( hipGetLastError() );
			}
		const int64_t total = ch * p.out_len;
		hipLaunchKernelGGL( k_any_overlap_add, dim3( (unsigned) ( ( total + 255 ) / 256 ) ), dim3( 256 ), 0, s, q );
		FLANHIP_CHECK( hipGetLastError() );
		rc = FLANHIP_OK;
		}
	else if( team_shape( lay.dft, W, lay.hop ) ) rc = run_synth_team( p, *plan, lay.dft, s );
	else if( sub_shape( lay.dft, W, lay.hop ) ) rc = run_synth_sub( p, lay.dft, s );
	else if( MrPlan mr_plan{}; mr_size( lay.dft, W, &mr_plan ) )
		{
		const size_t lds = mr_synth_lds( mr_plan.C, W, mr_plan.win_lds, mr_plan.kc_lds, mr_plan.ring_ws != 0 );
		FLANHIP_REQUIRE( !mr_plan.ring_ws || lay.mr_ring_offset != 0, FLANHIP_ERR_UNSUPPORTED, "the workspace holds no ring for this plan" );
		p.ring_ws = mr_plan.ring_ws ? reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.mr_ring_offset ) : nullptr;
		FLANHIP_REQUIRE( mr_plan_fits_kernels( mr_plan ), FLANHIP_ERR_UNSUPPORTED, "mixed-radix plan and kernels disagree" );
		auto kern = !mr_pingpong( mr_plan.C ) ? k_synthesize_mr<false, true> : mr_plan_is_big( mr_plan ) ? k_synthesize_mr<true, true> : k_synthesize_mr<true, false>;
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( kern, dim3( (unsigned) chains ), dim3( MR_THREADS ), lds, s, p, mr_plan );
		FLANHIP_CHECK( hipGetLastError() );
		rc = FLANHIP_OK;
		}
	else if( BigPlan big_plan{}; lay.big && big_size( lay.dft, W, &big_plan ) )
		{
		const int64_t blocks = big_blocks( chains, big_plan.P );
		FLANHIP_REQUIRE( blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "too many chains for one launch" );
		const bool ring_ws = lay.big_ring_offset != 0;
		const BigSynthExtra e{ reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.big_out_offset ), reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.big_head_offset ),
			ring_ws ? reinterpret_cast<float*>( reinterpret_cast<char*>( d_ws ) + lay.big_ring_offset ) : nullptr };
		const size_t lds = ring_ws ? big_analyze_lds( big_plan.C2 ) : big_synth_lds( big_plan.C2, W );
		const int bq = ( big_plan.C2 + MR_THREADS - 1 ) / MR_THREADS;
		auto kern = big_plan.mixed == 2 ? ( bq > 4 ? k_synthesize_big<8, 2> : bq > 2 ? k_synthesize_big<4, 2> : k_synthesize_big<2, 2> )
			: big_plan.mixed ? ( bq > 4 ? k_synthesize_big<8, 1> : bq > 2 ? k_synthesize_big<4, 1> : k_synthesize_big<2, 1> )
			: big_plan.C2 == 4096 ? k_synthesize_big<8> : big_plan.C2 == 2048 ? k_synthesize_big<4> : k_synthesize_big<2>;
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( kern, dim3( (unsigned) blocks ), dim3( MR_THREADS ), lds, s, p, big_plan, e );
		FLANHIP_CHECK( hipGetLastError() );
		// the units' streams and the chains' heads added up in a fixed order (no separate fix-up for these sizes)
		const int64_t total = ch * p.out_len;
		FLANHIP_REQUIRE( ( total + 255 ) / 256 < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "output too long for one launch" );
		if( total > 0 ) hipLaunchKernelGGL( k_big_reduce, dim3( (unsigned) ( ( total + 255 ) / 256 ) ), dim3( 256 ), 0, s, p, big_plan, e );
		FLANHIP_CHECK( hipGetLastError() );
		rc = FLANHIP_OK;
		}
	else if( BsPlan bs_plan{}; bs_size( lay.dft, W, &bs_plan ) )
		{
		FLANHIP_REQUIRE( plan->d_bs_tw && plan->d_bs_chirp && plan->d_bs_bh, FLANHIP_ERR_UNSUPPORTED, "chirp-z plan and tables disagree" );
		BsTables g{ plan->d_bs_tw, plan->d_bs_chirp, plan->d_bs_bh };
		if( bs_plan.glob )
			{
			FLANHIP_REQUIRE( lay.bsg_offset != 0, FLANHIP_ERR_UNSUPPORTED, "the workspace holds no chirp-z stretch for this plan" );
			g.scratch = reinterpret_cast<unsigned char*>( d_ws ) + lay.bsg_offset;
			g.scratch_stride = bsg_block_bytes( bs_plan.C, bs_plan.M, W );
			hipLaunchKernelGGL( ( k_synthesize_bs<true, false, true> ), dim3( (unsigned) chains ), dim3( MR_THREADS ), 0, s, p, bs_plan, g );
			}
		else
			{
			const size_t lds = bs_synth_lds( bs_plan.C, bs_plan.M, W, bs_plan.win_lds != 0 );
			auto kern = !bs_pingpong( bs_plan.M ) ? k_synthesize_bs<false, false> : bs_hoist( bs_plan, W ) ? k_synthesize_bs<true, true> : k_synthesize_bs<true, false>;
			FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kern ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
			hipLaunchKernelGGL( kern, dim3( (unsigned) chains ), dim3( MR_THREADS ), lds, s, p, bs_plan, g );
			}
		FLANHIP_CHECK( hipGetLastError() );
		rc = FLANHIP_OK;
		}
	else if( synth_fast_ok( lay.dft, W, lay.hop ) )
		{
		FastTables tb{ plan->d_tw1f, plan->d_tw3f, plan->d_tw2 };
		rc = v3_size( lay.dft ) ? run_synth_v3_hop( lay.dft, p, s ) : lay.dft == 2048 ? run_synth_fast_hop<10>( p, tb, s ) : run_synth_fast_hop<11>( p, tb, s );
		}
	else switch( ilog2( lay.dft ) - 1 )
		{
		case 4:  rc = run_synth<4, 8>( p, s ); break;
		case 5:  rc = run_synth<5, 8>( p, s ); break;
		case 6:  rc = run_synth<6, 8>( p, s ); break;
		case 7:  rc = run_synth<7, 8>( p, s ); break;
		case 8:  rc = run_synth<8, 8>( p, s ); break;
		case 9:  rc = run_synth<9, 1, 2>( p, s ); break;
		case 10: rc = run_synth<10, 1, 4>( p, s ); break;
		case 11: rc = run_synth<11, 1, 4>( p, s ); break;
		case 12: rc = run_synth<12, 1, kTeamWaves12>( p, s ); break;
		default: set_error( "unsupported dft size %d", lay.dft );
		}
	if( rc ) return rc;
	if( ( stages & 8 ) && p.head_len > 0 && p.chains_per_channel > 1 && !self_fix && !lay.big )
		{
		const bool quads = p.hop % 4 == 0 && ( W / 2 ) % 4 == 0 && p.head_len % 4 == 0 && p.out_len % 4 == 0
			&& ( reinterpret_cast<uintptr_t>( p.out ) & 15 ) == 0 && ( reinterpret_cast<uintptr_t>( p.head ) & 15 ) == 0;
		if( quads )
			{
			const int64_t threads = int64_t( ch ) * ( p.chains_per_channel - 1 ) * ( p.head_len / 4 );
			hipLaunchKernelGGL( k_ola_fixup4, dim3( (unsigned) ( ( threads + 255 ) / 256 ) ), dim3( 256 ), 0, s, p );
			}
		else hipLaunchKernelGGL( k_ola_fixup, dim3( (unsigned) chains ), dim3( 256 ), 0, s, p );
		FLANHIP_CHECK( hipGetLastError() );
		}
	return FLANHIP_OK;
	}

struct DeviceBuffer
	{
	void * p = nullptr;
	~DeviceBuffer() { if( p ) (void) hipFree( p ); }
	int alloc( size_t bytes ) { FLANHIP_CHECK( hipMalloc( &p, bytes ? bytes : 1 ) ); return FLANHIP_OK; }
	};

} // namespace flanhip

using namespace flanhip;

extern "C" {

int flanhip_analyze_dev( const float * d_audio, int64_t ch, int64_t n, float sr, int W, int hop, int dft, flanhip_MF * d_out, void * stream )
	{
	return launch_analyze( d_audio, ch, n, sr, W, hop, dft, d_out, nullptr, (hipStream_t) stream );
	}

int flanhip_analyze_dev_fused( const float * d_audio, int64_t ch, int64_t n, float sr, int W, int hop, int dft, flanhip_MF * d_out,
	void * d_synth_workspace, void * stream )
	{
	FLANHIP_REQUIRE( d_synth_workspace, FLANHIP_ERR_INVALID_ARG, "null workspace" );
	return launch_analyze( d_audio, ch, n, sr, W, hop, dft, d_out, d_synth_workspace, (hipStream_t) stream );
	}

int flanhip_synthesize_dev_fused( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W,
	float * d_out, void * d_ws, int * d_nan, void * stream )
	{
	return launch_synthesize( d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_nan, 1, (hipStream_t) stream );
	}

int flanhip_synthesize_prepass_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W, void * d_ws,
	double * d_total_out, int * d_nan, void * stream )
	{
	FLANHIP_REQUIRE( d_total_out, FLANHIP_ERR_INVALID_ARG, "null totals" );
	return launch_synthesize( d_pv, ch, F, bins, sr, ar, W, nullptr, d_ws, d_nan, 0, (hipStream_t) stream, nullptr, d_total_out, true );
	}

int flanhip_synthesize_dev_carry( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W, float * d_out,
	void * d_ws, const double * d_carry_in, int * d_nan, void * stream )
	{
	// the chain sums are in the workspace (flanhip_synthesize_prepass_dev): scan from the carry, synthesise
	return launch_synthesize( d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_nan, 3, (hipStream_t) stream, d_carry_in, nullptr, false );
	}

int flanhip_synthesize_dev_fused_checked( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W,
	float * d_out, void * d_ws, int * d_nan, void * stream )
	{
	return launch_synthesize( d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_nan, 2, (hipStream_t) stream );
	}

int flanhip_analyze( const float * audio, int64_t ch, int64_t n, float sr, int W, int hop, int dft,
	flanhip_MF * out, int64_t * num_pv_frames, volatile int * cancel )
	{
	FLANHIP_REQUIRE( audio && out, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	FLANHIP_REQUIRE( ch > 0 && n >= 0 && hop >= 1, FLANHIP_ERR_INVALID_ARG, "bad sizes" );
	if( int rc = require_device() ) return rc;
	const int64_t F = n / hop + 1;
	const int bins = dft / 2 + 1;
	if( num_pv_frames ) *num_pv_frames = F;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;       // flan_CANCEL_POINT, AudioPV.cpp:49
	DeviceBuffer d_audio, d_pv;
	if( int rc = d_audio.alloc( sizeof( float ) * size_t( ch ) * n ) ) return rc;
	if( int rc = d_pv.alloc( sizeof( flanhip_MF ) * size_t( ch ) * F * bins ) ) return rc;
	if( int rc_t = flanhip_upload( d_audio.p, audio, sizeof( float ) * size_t( ch ) * n ) ) return rc_t;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	if( int rc = launch_analyze( (const float*) d_audio.p, ch, n, sr, W, hop, dft, (flanhip_MF*) d_pv.p, nullptr, nullptr ) ) return rc;
	if( int rc = wait_cancellable( nullptr, cancel ? poll_volatile_int : nullptr, const_cast<int*>( cancel ) ) ) return rc;   // a flag raised while the kernels run stops them (core.hip)
	if( int rc_t = flanhip_download( out, d_pv.p, sizeof( flanhip_MF ) * size_t( ch ) * F * bins ) ) return rc_t;
	return FLANHIP_OK;
	}

int flanhip_synthesize_dev_stages( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W,
	float * d_out, void * d_ws, int * d_nan, int presummed, int stages, void * stream )
	{
	FLANHIP_REQUIRE( presummed >= 0 && presummed <= 2, FLANHIP_ERR_INVALID_ARG, "presummed: 0, 1 or 2" );
	return launch_synthesize( d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_nan, presummed, (hipStream_t) stream, nullptr, nullptr, false, stages );
	}

void flanhip_debug_option( int which, int value )
	{
	DebugOptions & o = debug_options();
	switch( which )
		{
		case FLANHIP_DEBUG_CHAIN_LEN:       o.chain_len = value; break;
		case FLANHIP_DEBUG_TARGET_CHAINS:   o.target_chains = value; break;
		case FLANHIP_DEBUG_FORCE_GENERIC:   o.force_generic = value; break;
		case FLANHIP_DEBUG_NO_FAST_DIV:     o.no_fast_div = value; break;
		case FLANHIP_DEBUG_ANA_VARIANT:     o.ana_variant = value; break;
		case FLANHIP_DEBUG_SYN_VARIANT:     o.syn_variant = value; break;
		case FLANHIP_DEBUG_ANA4096_OLD:     o.ana11_old = value; break;
		case FLANHIP_DEBUG_SYN4096_OLD:     o.syn11_old = value; break;
		case FLANHIP_DEBUG_RESAMPLE_DIRECT: o.resample_direct = value; break;
		case FLANHIP_DEBUG_FORCE_DIRECT:    o.force_direct = value; break;
		case FLANHIP_DEBUG_INLINE_FIXUP:    o.inline_fixup = value; break;
		case FLANHIP_DEBUG_WIDE_OFFSETS:    o.wide_offsets = value; break;
		case FLANHIP_DEBUG_NO_SUB:          o.no_sub = value; break;
		default: break;
		}
	}
#ifdef FLANHIP_STAMPS
// diagnostic build only: the per-section cycle sums of the stamped kernels (16 words; [15] = wavefronts that reported), then cleared
int flanhip_debug_read_stamps( unsigned long long * out16 )
	{
	FLANHIP_CHECK( hipDeviceSynchronize() );
	FLANHIP_CHECK( hipMemcpyFromSymbol( out16, HIP_SYMBOL( g_stamp_acc ), 16 * sizeof( unsigned long long ) ) );
	unsigned long long zero[16] = { 0 };
	FLANHIP_CHECK( hipMemcpyToSymbol( HIP_SYMBOL( g_stamp_acc ), zero, sizeof( zero ) ) );
	return FLANHIP_OK;
	}
// ... and the start / end of every wavefront's life in the last launch (s_memrealtime ticks of 10 ns; 2 x 4096 words)
int flanhip_debug_read_spans( unsigned long long * out8192 )
	{
	FLANHIP_CHECK( hipDeviceSynchronize() );
	FLANHIP_CHECK( hipMemcpyFromSymbol( out8192, HIP_SYMBOL( g_stamp_span ), 8192 * sizeof( unsigned long long ) ) );
	return FLANHIP_OK;
	}
#endif

size_t flanhip_synthesize_workspace_bytes( int64_t ch, int64_t F, int bins, float sr, float ar, int W )
	{
	SynthLayout lay;
	if( synth_layout( ch, F, bins, sr, ar, W, &lay ) ) return 0;
	size_t bytes = lay.total_bytes;
	// dft 1024 / 512: the A/B hook FLANHIP_DEBUG_ANA_VARIANT selects kernel configurations with other group sizes and chain counts, i.e. another
	// layout; a workspace sized under one setting of the hook holds every other's
	if( v3_size( lay.dft ) )
		{
		DebugOptions & o = debug_options();
		const int keep = o.ana_variant;
		for( int v = 0; v < 3; ++v )
			{
			o.ana_variant = v;
			SynthLayout alt;
			if( !synth_layout( ch, F, bins, sr, ar, W, &alt ) ) bytes = std::max( bytes, alt.total_bytes );
			}
		o.ana_variant = keep;
		}
	return bytes;
	}

int flanhip_synthesize_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W,
	float * d_out, void * d_ws, int * d_nan, void * stream )
	{
	return launch_synthesize( d_pv, ch, F, bins, sr, ar, W, d_out, d_ws, d_nan, false, (hipStream_t) stream );
	}

int flanhip_synthesize( const flanhip_MF * pv, int64_t ch, int64_t F, int bins, float sr, float ar, int W,
	float * out, int * nan_flag, volatile int * cancel )
	{
	FLANHIP_REQUIRE( pv && out, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	SynthLayout lay;
	if( int rc = synth_layout( ch, F, bins, sr, ar, W, &lay ) ) return rc;
	if( int rc = require_device() ) return rc;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;       // flan_CANCEL_POINT, AudioPV.cpp:115
	const size_t pv_bytes = sizeof( flanhip_MF ) * size_t( ch ) * F * bins;
	const size_t out_bytes = sizeof( float ) * size_t( ch ) * F * lay.hop;
	DeviceBuffer d_pv, d_out, d_ws, d_flag;
	if( int rc = d_pv.alloc( pv_bytes ) ) return rc;
	if( int rc = d_out.alloc( out_bytes ) ) return rc;
	if( int rc = d_ws.alloc( lay.total_bytes ) ) return rc;
	if( int rc = d_flag.alloc( sizeof( int ) ) ) return rc;
	FLANHIP_CHECK( hipMemset( d_flag.p, 0, sizeof( int ) ) );
	if( int rc_t = flanhip_upload( d_pv.p, pv, pv_bytes ) ) return rc_t;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	if( int rc = launch_synthesize( (const flanhip_MF*) d_pv.p, ch, F, bins, sr, ar, W, (float*) d_out.p, d_ws.p, (int*) d_flag.p, false, nullptr ) ) return rc;
	if( int rc = wait_cancellable( nullptr, cancel ? poll_volatile_int : nullptr, const_cast<int*>( cancel ) ) ) return rc;
	if( int rc_t = flanhip_download( out, d_out.p, out_bytes ) ) return rc_t;
	int flag = 0;
	FLANHIP_CHECK( hipMemcpy( &flag, d_flag.p, sizeof( int ), hipMemcpyDeviceToHost ) );
	if( nan_flag ) *nan_flag = flag;
	return FLANHIP_OK;
	}

} // extern "C"
