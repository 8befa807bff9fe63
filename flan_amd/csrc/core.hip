// core.hip -- error state, device plumbing, plan cache, shape helpers of the C ABI (include/flanhip.h).
#include "flanhip_internal.h"
#include "div_plans_proven.h"
#include "bs_plan.h"
#include <complex>
#include <memory>
#include <thread>
#include <chrono>
#include <atomic>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace flanhip {

static thread_local char g_error[512] = "";

void set_error( const char * fmt, ... )
	{
	va_list ap; va_start( ap, fmt );
	vsnprintf( g_error, sizeof( g_error ), fmt, ap );
	va_end( ap );
	}

int require_device()
	{
	int count = 0;
	if( hipGetDeviceCount( &count ) != hipSuccess || count <= 0 )
		{
		(void) hipGetLastError();
		set_error( "no HIP device visible: the phase-vocoder path has no CPU fallback" );
		return FLANHIP_ERR_NO_DEVICE;
		}
	return FLANHIP_OK;
	}

// WindowFunctions.cpp:8-13 as libstdc++ evaluates it: float product widened, ::cos(double), narrowed once.
static float hann_host( float x )
	{
	const float pi = std::acos( -1.0f );
	return float( 0.5f * ( 1.0f - std::cos( double( 2.0f * pi * x ) ) ) );
	}

// The plan cache.  Plans of the power-of-two sizes with FFT kernels (32 ... 8192: a few dozen KB each, a closed set of shapes per window) stay for
// the life of the process, like FFTW's wisdom; plans of any OTHER dft size (the mixed-radix and direct-sum paths: tables of 8 bytes x dft and
// more, up to 2^20) live in a small least-recently-used set -- a caller that sweeps sizes no longer grows the device's memory without bound.
// A plan is handed out as a shared reference: the caller keeps it until its kernels are launched, and an evicted plan's tables are freed when
// its last holder lets go (hipFree waits for the device, so kernels already launched finish first).  Tables are worked out BEFORE the lock is
// taken: a first use of a large size does not stall other threads' launches behind a hundred thousand cos / sin calls.
PlanRef::~PlanRef()
	{
	(void) hipFree( plan.d_window ); (void) hipFree( plan.d_tw ); (void) hipFree( plan.d_tw2 ); (void) hipFree( plan.d_tw1f ); (void) hipFree( plan.d_tw3f );
	(void) hipFree( plan.d_bs_tw ); (void) hipFree( plan.d_bs_chirp ); (void) hipFree( plan.d_bs_bh ); (void) hipFree( plan.d_team ); (void) hipFree( plan.d_dump );
	(void) hipGetLastError();
	}
static std::mutex g_plan_mutex;
typedef std::tuple<int, int, int> PlanKey;
// (heap objects that are never destroyed: at process exit the HIP runtime may be gone before static destructors run)
static std::map<PlanKey, std::shared_ptr<const PlanRef>> & plan_map() { static auto * m = new std::map<PlanKey, std::shared_ptr<const PlanRef>>; return *m; }
static std::vector<PlanKey> & plan_lru() { static auto * v = new std::vector<PlanKey>; return *v; }   // the evictable plans, most recently used last
static constexpr size_t kEvictablePlans = 8;            // per device
static bool plan_is_permanent( int dft_size ) { return is_pow2( dft_size ) && dft_size >= 32 && dft_size <= 8192; }

// exp( -2 pi i num / den ) rounded to float from an octant-reduced evaluation in long double: exact at the quarter turns and exactly symmetric
// ( the value at pi - a is the value at a with its real part negated, ... ), which the team kernels rely on where one bin is reached twice
static cf unit_minus_exact( int64_t num, int64_t den )
	{
	const int64_t n = ( ( num % den ) + den ) % den;
	const int64_t o = ( 8 * n ) / den, r = 8 * n - o * den;                       // octant, and the rest in 1 / ( 8 den ) turns
	const long double two_pi = 6.283185307179586476925286766559005768L;
	const long double a = two_pi * ( long double )( ( o & 1 ) ? den - r : r ) / ( long double )( 8 * den );   // in [0, pi / 4]
	const float c = float( cosl( a ) ), s = float( sinl( a ) );
	float x, y;                                                                     // ( cos, sin ) of 2 pi n / den
	switch( int( o ) )
		{
		case 0: x = c; y = s; break;
		case 1: x = s; y = c; break;
		case 2: x = -s; y = c; break;
		case 3: x = -c; y = s; break;
		case 4: x = -c; y = -s; break;
		case 5: x = -s; y = -c; break;
		case 6: x = s; y = -c; break;
		default: x = c; y = -s; break;
		}
	return cf{ x, y == 0.0f ? 0.0f : -y };
	}

int get_plan( int window_size, int dft_size, std::shared_ptr<const PlanRef> * out )
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	const PlanKey key = std::make_tuple( device, window_size, dft_size );
	auto touch = [&]()
		{
		auto & lru = plan_lru();
		for( size_t i = 0; i < lru.size(); ++i ) if( lru[i] == key ) { lru.erase( lru.begin() + long( i ) ); lru.push_back( key ); break; }
		};
		{
		std::lock_guard<std::mutex> lock( g_plan_mutex );
		auto it = plan_map().find( key );
		if( it != plan_map().end() ) { touch(); *out = it->second; return FLANHIP_OK; }
		}

	const int C = dft_size / 2;
	std::vector<float> win( window_size );
	for( int i = 0; i < window_size; ++i ) win[i] = hann_host( float( i ) / float( window_size - 1 ) );  // AudioPV.cpp:30-34
	std::vector<cf> tw( C ), tw2( C + 1 );
	const double pi = 3.14159265358979323846;
	for( int k = 0; k < C; ++k ) tw[k] = cf{ float( std::cos( -2.0 * pi * k / C ) ), float( std::sin( -2.0 * pi * k / C ) ) };
	for( int k = 0; k <= C; ++k ) tw2[k] = cf{ float( std::cos( -pi * k / C ) ), float( std::sin( -pi * k / C ) ) };

	auto ref = std::make_shared<PlanRef>();
	Plan & plan = ref->plan;
	FLANHIP_CHECK( hipMalloc( &plan.d_window, sizeof( float ) * window_size ) );
	FLANHIP_CHECK( hipMalloc( &plan.d_tw, sizeof( cf ) * C ) );
	FLANHIP_CHECK( hipMalloc( &plan.d_tw2, sizeof( cf ) * ( C + 1 ) ) );
	FLANHIP_CHECK( hipMemcpy( plan.d_window, win.data(), sizeof( float ) * window_size, hipMemcpyHostToDevice ) );
	FLANHIP_CHECK( hipMemcpy( plan.d_tw, tw.data(), sizeof( cf ) * C, hipMemcpyHostToDevice ) );
	FLANHIP_CHECK( hipMemcpy( plan.d_tw2, tw2.data(), sizeof( cf ) * ( C + 1 ), hipMemcpyHostToDevice ) );
	FLANHIP_CHECK( hipMalloc( &plan.d_dump, 1024 ) );
	if( dft_size == 2048 || dft_size == 4096 )
		{
		const int R3 = C / 256;
		std::vector<cf> tw1( 15 * 16 ), tw3( size_t( R3 - 1 ) * 256 );
		for( int r = 1; r < 16; ++r ) for( int k = 0; k < 16; ++k )
			tw1[( r - 1 ) * 16 + k] = cf{ float( std::cos( -2.0 * pi * r * k / 256.0 ) ), float( std::sin( -2.0 * pi * r * k / 256.0 ) ) };
		for( int r = 1; r < R3; ++r ) for( int j = 0; j < 256; ++j )
			tw3[size_t( r - 1 ) * 256 + j] = cf{ float( std::cos( -2.0 * pi * r * j / C ) ), float( std::sin( -2.0 * pi * r * j / C ) ) };
		FLANHIP_CHECK( hipMalloc( &plan.d_tw1f, sizeof( cf ) * tw1.size() ) );
		FLANHIP_CHECK( hipMalloc( &plan.d_tw3f, sizeof( cf ) * tw3.size() ) );
		FLANHIP_CHECK( hipMemcpy( plan.d_tw1f, tw1.data(), sizeof( cf ) * tw1.size(), hipMemcpyHostToDevice ) );
		FLANHIP_CHECK( hipMemcpy( plan.d_tw3f, tw3.data(), sizeof( cf ) * tw3.size(), hipMemcpyHostToDevice ) );
		}
	if( const int R = team_radix( dft_size ) )
		{
		// the team kernels' tables (pv_kernels_team.h)
		const TeamTableLayout l = team_table_layout( R );
		const int64_t CT = int64_t( 1024 ) * R;
		std::vector<cf> t( size_t( l.total ) );
		for( int r = 1; r < 16; ++r ) for( int k = 0; k < 16; ++k ) t[size_t( l.tw1 + ( r - 1 ) * 16 + k )] = unit_minus_exact( int64_t( r ) * k, 256 );
		for( int r = 1; r < 4; ++r ) for( int j = 0; j < 256; ++j ) t[size_t( l.tw3 + ( r - 1 ) * 256 + j )] = unit_minus_exact( int64_t( r ) * j, 1024 );
		for( int r = 1; r < R; ++r ) for( int k = 0; k < 512; ++k ) t[size_t( l.twj + ( r - 1 ) * 512 + k )] = unit_minus_exact( int64_t( r ) * k, CT );
		for( int j = 0; j < R / 2; ++j ) for( int k = 0; k < 512; ++k ) t[size_t( l.tws + j * 512 + k )] = unit_minus_exact( k + 1024 * int64_t( j ), 2 * CT );
		for( int j = 0; j < R; ++j ) t[size_t( l.two + j )] = unit_minus_exact( 512 + 1024 * int64_t( j ), 2 * CT );
		for( int r = 0; r < R; ++r ) t[size_t( l.two + R + r )] = unit_minus_exact( 512 * int64_t( r ), CT );
		FLANHIP_CHECK( hipMalloc( &plan.d_team, sizeof( cf ) * t.size() ) );
		FLANHIP_CHECK( hipMemcpy( plan.d_team, t.data(), sizeof( cf ) * t.size(), hipMemcpyHostToDevice ) );
		}
	if( BsPlan bp{}; bs_plan_in_use( dft_size, &bp ) )
		{
		// Bluestein's tables (pv_kernels_bs.h), all evaluated in double: the chirp from n^2 mod 2 C (exact integers), its transform by a radix-2 FFT
		const int M = bp.M;
		std::vector<cf> twm( M ); std::vector<d2> chirp( C ), bh( M );
		for( int j = 0; j < M; ++j ) twm[j] = cf{ float( std::cos( -2.0 * pi * j / M ) ), float( std::sin( -2.0 * pi * j / M ) ) };
		std::vector<std::complex<double>> b( M, std::complex<double>( 0.0, 0.0 ) );
		for( int n = 0; n < C; ++n )
			{
			const int64_t q = ( int64_t( n ) * n ) % ( 2 * int64_t( C ) );
			const std::complex<double> w( std::cos( pi * double( q ) / C ), std::sin( pi * double( q ) / C ) );
			chirp[n] = d2{ w.real(), w.imag() };
			b[n] = w;
			if( n ) b[M - n] = w;
			}
		for( int i = 1, j = 0; i < M; ++i )                                          // bit reversal
			{
			int bit = M >> 1;
			for( ; j & bit; bit >>= 1 ) j ^= bit;
			j ^= bit;
			if( i < j ) std::swap( b[i], b[j] );
			}
		for( int len = 2; len <= M; len <<= 1 )
			for( int i = 0; i < M; i += len )
				for( int k = 0; k < len / 2; ++k )
					{
					const std::complex<double> w( std::cos( -2.0 * pi * k / len ), std::sin( -2.0 * pi * k / len ) );
					const std::complex<double> u = b[i + k], v = b[i + k + len / 2] * w;
					b[i + k] = u + v; b[i + k + len / 2] = u - v;
					}
		for( int k = 0; k < M; ++k ) bh[k] = d2{ b[k].real() / M, b[k].imag() / M };
		FLANHIP_CHECK( hipMalloc( &plan.d_bs_tw, sizeof( cf ) * M ) );
		FLANHIP_CHECK( hipMalloc( &plan.d_bs_chirp, sizeof( d2 ) * C ) );
		FLANHIP_CHECK( hipMalloc( &plan.d_bs_bh, sizeof( d2 ) * M ) );
		FLANHIP_CHECK( hipMemcpy( plan.d_bs_tw, twm.data(), sizeof( cf ) * M, hipMemcpyHostToDevice ) );
		FLANHIP_CHECK( hipMemcpy( plan.d_bs_chirp, chirp.data(), sizeof( d2 ) * C, hipMemcpyHostToDevice ) );
		FLANHIP_CHECK( hipMemcpy( plan.d_bs_bh, bh.data(), sizeof( d2 ) * M, hipMemcpyHostToDevice ) );
		}
	std::shared_ptr<const PlanRef> evicted;                                         // (released outside the lock: freeing waits for the device)
		{
		std::lock_guard<std::mutex> lock( g_plan_mutex );
		auto it = plan_map().find( key );
		if( it != plan_map().end() ) { touch(); *out = it->second; return FLANHIP_OK; }   // another thread was faster: ours is dropped
		plan_map()[key] = ref;
		if( !plan_is_permanent( dft_size ) )
			{
			auto & lru = plan_lru();
			lru.push_back( key );
			// the bound is per DEVICE: a process that drives eight GPUs with a couple of such sizes each must not evict on every call
			size_t mine = 0;
			for( const PlanKey & k : lru ) mine += std::get<0>( k ) == device;
			if( mine > kEvictablePlans )
				for( size_t i = 0; i < lru.size(); ++i ) if( std::get<0>( lru[i] ) == device )
					{
					auto old = plan_map().find( lru[i] );
					if( old != plan_map().end() ) { evicted = old->second; plan_map().erase( old ); }
					lru.erase( lru.begin() + long( i ) );
					break;
					}
			}
		}
	*out = ref;
	return FLANHIP_OK;
	}

// every float bit pattern: does fma( fma( -x*rc, c, x ), rc, x*rc ) equal the hardware (IEEE) quotient x / c ?
__global__ __launch_bounds__( 256 ) void k_verify_div( float c, float rc, unsigned int * mismatches )
	{
	unsigned int bad = 0;
	for( uint64_t u = uint64_t( blockIdx.x ) * blockDim.x + threadIdx.x; u < ( uint64_t( 1 ) << 32 ); u += uint64_t( gridDim.x ) * blockDim.x )
		{
		const float x = __uint_as_float( uint32_t( u ) );
		if( !( __builtin_fabsf( x ) >= 1.0e-30f ) || !( __builtin_fabsf( x ) <= 3.4028235e38f ) ) continue;
		const float q0 = x * rc;
		const float q = __builtin_fmaf( __builtin_fmaf( -q0, c, x ), rc, q0 );
		const float ref = x / c;
		bad += ( __float_as_uint( q ) != __float_as_uint( ref ) ) && !( ref != ref );
		}
	if( bad ) atomicAdd( mismatches, bad );
	}

static std::map<uint32_t, DivPlan> g_div_plans;

int get_div_plan( float c, DivPlan * out )
	{
	uint32_t key; std::memcpy( &key, &c, 4 );
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	// the calling thread's FLANHIP_DEBUG_NO_FAST_DIV hook is applied to what is handed out, never to what is cached: the cache is the process's,
	// the hook the thread's
	const bool no_fast = debug_options().no_fast_div != 0;
	std::lock_guard<std::mutex> lock( g_plan_mutex );
	auto it = g_div_plans.find( key );
	if( it != g_div_plans.end() ) { *out = it->second; if( no_fast ) out->exact = 0; return FLANHIP_OK; }
	DivPlan d{ c, 1.0f / c, 0 };
	// the analysis rates of everyday sample rates and hops were tried offline, all 2^32 dividends each (tools/check_div_c.cpp ->
	// div_plans_proven.h): no launch of k_verify_div (2.4 ms) for those
	bool known = false;
	for( const ProvenDiv & pd : kProvenDivs ) if( pd.bits == key ) { d.exact = pd.exact; known = true; break; }
	if( !known && no_fast ) { *out = d; return FLANHIP_OK; }                      // (nothing proven, nothing cached: the hardware division it is)
	if( !known && c > 1.0e-10f && c < 1.0e10f )
		{
		unsigned int * d_bad = nullptr, bad = 1;
		FLANHIP_CHECK( hipMalloc( &d_bad, sizeof( unsigned int ) ) );
		FLANHIP_CHECK( hipMemset( d_bad, 0, sizeof( unsigned int ) ) );
		hipLaunchKernelGGL( k_verify_div, dim3( 4096 ), dim3( 256 ), 0, nullptr, d.c, d.rc, d_bad );
		FLANHIP_CHECK( hipGetLastError() );
		FLANHIP_CHECK( hipMemcpy( &bad, d_bad, sizeof( unsigned int ), hipMemcpyDeviceToHost ) );
		FLANHIP_CHECK( hipFree( d_bad ) );
		d.exact = bad == 0;
		}
	g_div_plans[key] = d;
	*out = d;
	if( no_fast ) out->exact = 0;
	return FLANHIP_OK;
	}

// Test / A-B hooks (flanhip_debug_option): PER CALLING THREAD, all off by default, never set by the product paths (the C++ classes, bench.py's
// timed region).  They replace what used to be process-wide variables and getenv() calls on every entry: a thread that sets one changes its own
// calls only, and the library reads no environment variable on its call paths.
DebugOptions & debug_options()
	{
	static thread_local DebugOptions o;
	return o;
	}

bool force_generic() { return debug_options().force_generic != 0; }

// compute units of the current device (hipDeviceProp_t::multiProcessorCount), cached per device: what "the wavefronts the chip holds at once"
// is counted from -- 256 on a whole MI355X, 32 per XCD-partition in CPX mode
int cu_count()
	{
	static std::atomic<int> cached[64];
	int device = 0;
	if( hipGetDevice( &device ) != hipSuccess ) { (void) hipGetLastError(); return 256; }
	if( device >= 0 && device < 64 && cached[device].load( std::memory_order_relaxed ) > 0 ) return cached[device].load( std::memory_order_relaxed );
	int n = 0;
	if( hipDeviceGetAttribute( &n, hipDeviceAttributeMultiprocessorCount, device ) != hipSuccess || n <= 0 ) { (void) hipGetLastError(); n = 256; }
	if( device >= 0 && device < 64 ) cached[device].store( n, std::memory_order_relaxed );
	return n;
	}

int choose_chain_length( int64_t num_channels, int64_t num_frames, int min_len, int target_chains, int group )
	{
	if( const int v = debug_options().chain_len ) { if( v > 0 ) return std::max( v, min_len ); }     // (tests: results must not depend on the cut)
	// One wavefront per chain.  target_chains = the wavefronts the chip holds at once for this kernel.  All chains run
	// concurrently and cost the same (L frames + one halo frame / overlap head), so the launch takes as long as ONE chain:
	// the best cut is the longest chain that still gives every resident wavefront slot a chain -- a single round of blocks,
	// the halo amortised over as many frames as possible.  Chains never span channels, so channels * ceil(F / L) can exceed
	// the slots by a few chains when the channel count does not divide them; one chain too many would start a second,
	// nearly empty round and double the time, so L grows until the grid fits.
	// Very long inputs are cut into R full rounds of chains of at most ~512 frames (the halo is then 0.2 % anyway and each
	// round works on a compact part of the buffers).
	const int64_t total = num_channels * num_frames;
	const int64_t cap = 512;
	int64_t L = ( total + target_chains - 1 ) / target_chains;
	const int64_t rounds = std::max<int64_t>( ( L + cap - 1 ) / cap, 1 );
	const int64_t slots = int64_t( target_chains ) * rounds;
	L = ( total + slots - 1 ) / slots;
	if( L < 4 ) L = 4;
	if( L < min_len ) L = min_len;
	while( num_channels * ( ( num_frames + L - 1 ) / L ) > slots && L < num_frames ) ++L;
	if( L > ( int64_t( 1 ) << 30 ) ) L = int64_t( 1 ) << 30;
	// A block of the tuned kernels is a GROUP of `group` consecutive chains of ONE channel (8 wavefronts at dft 2048, 4 teams at dft 4096, 8 ... 32 chains of the
	// dft 512 ... 128 kernels): with fewer chains per channel than that its other wavefronts idle -- 1024 channels of half a second each are 2 chains per channel at
	// the cut above, blocks of 8 wavefronts with 2 at work, four rounds of them: 1.13 ms where 8 ch x 60 s, the same frames, take 0.27 (round 6,
	// tools/input_sweep.py).  So: shorter chains that fill whole groups where that costs fewer block rounds x frames.  Cost of a cut: rounds of resident blocks x
	// ( L + 2 frame slots for the halo frame and a chain's fixed part ); only a cut that is 10 % cheaper replaces the one above.
	if( group > 1 )
		{
		const int64_t bslots = std::max<int64_t>( target_chains / group, 1 );
		auto cost = [&]( int64_t len )
			{
			const int64_t cpc = ( num_frames + len - 1 ) / len, blocks = num_channels * ( ( cpc + group - 1 ) / group );
			return ( ( blocks + bslots - 1 ) / bslots ) * ( len + 2 );
			};
		// (candidates: whole groups per channel, from one group up -- longer chains than the cut above too: 24 channels x 20 s are 85 chains per channel there, 11 blocks
		// per channel, 264 blocks for 256 CUs: a second round for eight blocks; 80 chains per channel fit one.  tools/channel_sweep.py)
		const int64_t first_cost = cost( L );
		int64_t best = L, best_cost = first_cost;
		for( int64_t m = 1; m <= 4096; ++m )
			{
			const int64_t cpc = m * group;
			if( cpc > num_frames ) break;
			const int64_t len = ( num_frames + cpc - 1 ) / cpc;
			if( len < std::max<int64_t>( min_len, 4 ) ) break;
			if( len > cap && rounds > 1 ) continue;                                  // (very long inputs keep their rounds of chains of at most ~512 frames)
			const int64_t c = cost( len );
			if( c * 10 < first_cost * 9 && c < best_cost ) { best = len; best_cost = c; }
			}
		L = best;
		}
	return int( L );
	}

static std::mutex g_ws_mutex;
static std::map<const void*, std::pair<int, int>> g_ws_producer;
void note_workspace_producer( const void * d_ws, int kind, int epoch )
	{
	std::lock_guard<std::mutex> lock( g_ws_mutex );
	if( !kind ) { g_ws_producer.erase( d_ws ); return; }
	// addresses the library never sees freed (a caller's own allocator) would pile up: forgetting everything is always safe -- a workspace
	// without a note gets the scan kernel
	if( g_ws_producer.size() >= 1024 ) g_ws_producer.clear();
	g_ws_producer[d_ws] = std::make_pair( kind, epoch );
	}
int workspace_producer( const void * d_ws, int * epoch )
	{
	std::lock_guard<std::mutex> lock( g_ws_mutex );
	auto it = g_ws_producer.find( d_ws );
	if( epoch ) *epoch = it == g_ws_producer.end() ? 0 : it->second.second;
	return it == g_ws_producer.end() ? 0 : it->second.first;
	}

// ---- cancellation inside a launch (defines.h:49-62: the reference polls its flag once per frame, AudioPV.cpp:49,115) -----------------
// Every host thread holds, per device, a block of CANCEL WORDS in fine-grained device memory (coherent across the chip's XCDs and with the copy
// engines; page-locked coherent host memory if that cannot be had): one word per STREAM the thread launches conversions on, each on a cache
// line of its own.  A conversion kernel gets the address of its stream's word and reads it, past the caches, when a block starts (the
// direct-sum kernels also every few batches of frames).  wait_cancellable( s ) waits for stream s while it polls the caller's flag; when that
// rises it sets THAT stream's word from a side stream, the blocks still to start retire at once (a block of the FFT kernels walks at most 512
// frames: ~3 ms), and the call returns FLANHIP_ERR_CANCELLED.  The scope of a cancellation is therefore the wait's own stream: what the
// thread has in flight on its other streams, and what other threads have in flight anywhere, is not touched (round 3 had one word per thread:
// a wait on stream A also stopped the thread's kernels on stream B, whose later plain synchronisation then reported success on half-written
// output).  A thread with more than kCancelSlots live streams on one device re-uses the least recently used slot.
// The blocks are drawn from a process-wide pool and go back to it when their thread ends (no HIP call in a thread's destructor, no
// allocation or stream per short-lived thread: the pool never holds more blocks than threads were alive at once).
constexpr int kCancelSlots = 64, kCancelStride = 16;                                // words; one 64-byte line per slot
struct CancelBlock
	{
	int device = -1;
	int * dev = nullptr; volatile int * host = nullptr;                               // host != nullptr: the host-memory fallback
	hipStream_t side = nullptr;
	hipStream_t owner[kCancelSlots] = {};                                             // the streams this thread has launched on, most recently used first ...
	int slot_of[kCancelSlots] = {};                                                   // ... and the slot each of them holds
	int used = 0;
	};
static std::mutex g_cancel_pool_mutex;
static std::vector<CancelBlock*> g_cancel_pool;                                     // idle blocks (their words are 0: a wait clears what it raised)
struct ThreadCancel
	{
	std::vector<CancelBlock*> blocks;                                                 // one per device this thread has used
	~ThreadCancel()
		{
		std::lock_guard<std::mutex> lock( g_cancel_pool_mutex );
		for( CancelBlock * b : blocks ) { b->used = 0; g_cancel_pool.push_back( b ); }
		}
	};
static thread_local ThreadCancel t_cancel;

static CancelBlock * thread_cancel_block()
	{
	int device = 0;
	if( hipGetDevice( &device ) != hipSuccess ) { (void) hipGetLastError(); return nullptr; }
	for( CancelBlock * b : t_cancel.blocks ) if( b->device == device ) return b;
		{
		std::lock_guard<std::mutex> lock( g_cancel_pool_mutex );
		for( size_t i = 0; i < g_cancel_pool.size(); ++i )
			if( g_cancel_pool[i]->device == device )
				{
				CancelBlock * b = g_cancel_pool[i];
				g_cancel_pool.erase( g_cancel_pool.begin() + long( i ) );
				t_cancel.blocks.push_back( b );
				return b;
				}
		}
	CancelBlock * b = new CancelBlock;
	b->device = device;
	const size_t bytes = sizeof( int ) * kCancelSlots * kCancelStride;
	void * d = nullptr;
	if( hipExtMallocWithFlags( &d, bytes, hipDeviceMallocFinegrained ) == hipSuccess && hipMemset( d, 0, bytes ) == hipSuccess )
		{
		b->dev = static_cast<int*>( d );
		if( hipStreamCreateWithFlags( &b->side, hipStreamNonBlocking ) != hipSuccess ) { (void) hipGetLastError(); (void) hipFree( d ); delete b; return nullptr; }
		}
	else
		{
		(void) hipGetLastError();
		void * h = nullptr; void * dp = nullptr;
		if( hipHostMalloc( &h, bytes, hipHostMallocCoherent | hipHostMallocMapped ) != hipSuccess ) { (void) hipGetLastError(); delete b; return nullptr; }
		std::memset( h, 0, bytes );
		if( hipHostGetDevicePointer( &dp, h, 0 ) != hipSuccess ) { (void) hipGetLastError(); (void) hipHostFree( h ); delete b; return nullptr; }
		b->dev = static_cast<int*>( dp ); b->host = static_cast<volatile int*>( h );
		}
	t_cancel.blocks.push_back( b );
	return b;
	}

// the slot of stream s in the thread's block.  owner[] / slot_of[] are ordered most recently used first; a new stream takes a slot nobody
// has had yet, or -- with kCancelSlots streams live -- the least recently used stream's
static int cancel_slot( CancelBlock & b, hipStream_t s )
	{
	int at = -1;
	for( int i = 0; i < b.used; ++i ) if( b.owner[i] == s ) { at = i; break; }
	int slot;
	if( at >= 0 ) slot = b.slot_of[at];
	else if( b.used < kCancelSlots ) { at = b.used++; slot = at; }
	else { at = kCancelSlots - 1; slot = b.slot_of[at]; }
	for( int i = at; i > 0; --i ) { b.owner[i] = b.owner[i - 1]; b.slot_of[i] = b.slot_of[i - 1]; }
	b.owner[0] = s; b.slot_of[0] = slot;
	return slot;
	}

int * thread_cancel_word( hipStream_t s )
	{
	CancelBlock * b = thread_cancel_block();
	return b ? b->dev + cancel_slot( *b, s ) * kCancelStride : nullptr;
	}

static void set_cancel_word( CancelBlock & b, int slot, int value, bool wait )
	{
	if( b.host ) { b.host[slot * kCancelStride] = value; std::atomic_thread_fence( std::memory_order_seq_cst ); return; }
	(void) hipMemsetAsync( b.dev + slot * kCancelStride, value, sizeof( int ), b.side );   // (a byte value: 1 gives 0x01010101 -- non-zero is all that counts)
	if( wait ) (void) hipStreamSynchronize( b.side );
	}

int wait_cancellable( hipStream_t s, int ( *poll )( void * ), void * user )
	{
	CancelBlock * b = poll ? thread_cancel_block() : nullptr;
	if( !b ) { FLANHIP_CHECK( hipStreamSynchronize( s ) ); return ( poll && poll( user ) ) ? FLANHIP_ERR_CANCELLED : FLANHIP_OK; }
	const int slot = cancel_slot( *b, s );
	// most waits end within the first few hundred microseconds and never see the flag rise: poll the stream itself for that long (no event, no
	// sleep quantum on top of a 70 us round trip), then settle into event queries with short sleeps
	for( int spins = 0; spins < 2000; ++spins )
		{
		const hipError_t q = hipStreamQuery( s );
		if( q == hipSuccess ) return poll( user ) ? FLANHIP_ERR_CANCELLED : FLANHIP_OK;
		if( q != hipErrorNotReady ) { (void) hipGetLastError(); set_error( "hipStreamQuery failed" ); return FLANHIP_ERR_HIP; }
		if( poll( user ) ) break;
		std::this_thread::yield();
		}
	hipEvent_t done;
	FLANHIP_CHECK( hipEventCreateWithFlags( &done, hipEventDisableTiming ) );
	if( hipEventRecord( done, s ) != hipSuccess ) { (void) hipEventDestroy( done ); set_error( "hipEventRecord failed" ); return FLANHIP_ERR_HIP; }
	bool raised = false;
	for( ;; )
		{
		const hipError_t q = hipEventQuery( done );
		if( q == hipSuccess ) break;
		if( q != hipErrorNotReady ) { (void) hipEventDestroy( done ); if( raised ) set_cancel_word( *b, slot, 0, true ); set_error( "hipEventQuery failed" ); return FLANHIP_ERR_HIP; }
		if( !raised && poll( user ) ) { set_cancel_word( *b, slot, 1, false ); raised = true; }
		std::this_thread::sleep_for( std::chrono::microseconds( 50 ) );
		}
	(void) hipEventDestroy( done );
	if( raised ) { set_cancel_word( *b, slot, 0, true ); return FLANHIP_ERR_CANCELLED; }   // (the stream has drained: nobody reads the word now)
	return poll( user ) ? FLANHIP_ERR_CANCELLED : FLANHIP_OK;
	}

int next_epoch()
	{
	// 1, 2, 3, ... 2^31 - 1, 1, ...: never 0, and two successive producer launches never share a number (the workspace words are
	// recycled uncleared through the device cache: a repeated number would make stale words look freshly set)
	static std::atomic<uint32_t> epoch{ 0 };
	return int( epoch.fetch_add( 1 ) % 0x7fffffffu ) + 1;
	}

} // namespace flanhip

using namespace flanhip;

extern "C" {

int flanhip_version( void ) { return 100; }   // 0.1.0
const char * flanhip_last_error( void ) { return g_error; }

int flanhip_device_count( void )
	{
	int count = 0;
	if( hipGetDeviceCount( &count ) != hipSuccess ) { (void) hipGetLastError(); return 0; }
	return count;
	}

int flanhip_set_device( int device )
	{
	if( int rc = require_device() ) return rc;
	FLANHIP_CHECK( hipSetDevice( device ) );
	return FLANHIP_OK;
	}

int flanhip_get_device( int * device )
	{
	FLANHIP_REQUIRE( device, FLANHIP_ERR_INVALID_ARG, "null out pointer" );
	if( int rc = require_device() ) return rc;
	FLANHIP_CHECK( hipGetDevice( device ) );
	return FLANHIP_OK;
	}

int64_t flanhip_num_pv_frames( int64_t num_audio_frames, int hop )
	{
	if( hop <= 0 || num_audio_frames < 0 ) return -1;
	return num_audio_frames / hop + 1;                 // AudioPV.cpp:17: std::ceil of an INTEGER quotient, + 1
	}

int flanhip_hop_size( float sample_rate, float analysis_rate )
	{
	return int( sample_rate / analysis_rate );         // PVBuffer.cpp:381-384
	}

int64_t flanhip_modify_time_out_frames( const float * mod, int64_t num_frames, int num_bins, float sample_rate, int hop )
	{
	if( !mod || num_frames <= 0 || num_bins <= 0 || hop <= 0 ) return -1;
	float mx = mod[0];                                  // FunctionSample::maximum(), FunctionSample.h:156-160
	for( int64_t i = 1; i < num_frames * num_bins; ++i ) mx = std::max( mx, mod[i] );
	const float last = std::ceil( mx * float( sample_rate ) / float( hop ) );  // PVModify.cpp:312, PVBuffer.cpp:428-431
	return int64_t( int32_t( last ) );                  // :315 float -> Frame
	}

int flanhip_malloc( void ** dptr, size_t bytes )
	{
	FLANHIP_REQUIRE( dptr, FLANHIP_ERR_INVALID_ARG, "null out pointer" );
	if( int rc = require_device() ) return rc;
	FLANHIP_CHECK( hipMalloc( dptr, bytes ? bytes : 1 ) );
	return FLANHIP_OK;
	}

int flanhip_free( void * dptr )
	{
	if( !dptr ) return FLANHIP_OK;
	note_workspace_producer( dptr, 0 );                                               // (a freed address may come back as somebody else's workspace)
	FLANHIP_CHECK( hipFree( dptr ) );
	return FLANHIP_OK;
	}

int flanhip_host_malloc( void ** hptr, size_t bytes )
	{
	FLANHIP_REQUIRE( hptr, FLANHIP_ERR_INVALID_ARG, "null out pointer" );
	if( int rc = require_device() ) return rc;
	FLANHIP_CHECK( hipHostMalloc( hptr, bytes ? bytes : 1, hipHostMallocDefault ) );
	return FLANHIP_OK;
	}

int flanhip_host_free( void * hptr )
	{
	if( !hptr ) return FLANHIP_OK;
	FLANHIP_CHECK( hipHostFree( hptr ) );
	return FLANHIP_OK;
	}

int flanhip_memcpy_h2d( void * dst, const void * src, size_t bytes, void * stream )
	{
	FLANHIP_CHECK( hipMemcpyAsync( dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t) stream ) );
	return FLANHIP_OK;
	}

int flanhip_memcpy_d2h( void * dst, const void * src, size_t bytes, void * stream )
	{
	FLANHIP_CHECK( hipMemcpyAsync( dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream ) );
	return FLANHIP_OK;
	}

int flanhip_memset( void * dst, int value, size_t bytes, void * stream )
	{
	FLANHIP_CHECK( hipMemsetAsync( dst, value, bytes, (hipStream_t) stream ) );
	return FLANHIP_OK;
	}

int flanhip_stream_create( void ** stream )
	{
	FLANHIP_REQUIRE( stream, FLANHIP_ERR_INVALID_ARG, "null out pointer" );
	if( int rc = require_device() ) return rc;
	hipStream_t s = nullptr;
	FLANHIP_CHECK( hipStreamCreateWithFlags( &s, hipStreamNonBlocking ) );
	*stream = s;
	return FLANHIP_OK;
	}

int flanhip_stream_destroy( void * stream )
	{
	if( !stream ) return FLANHIP_OK;
	FLANHIP_CHECK( hipStreamDestroy( (hipStream_t) stream ) );
	return FLANHIP_OK;
	}

int flanhip_stream_synchronize( void * stream )
	{
	FLANHIP_CHECK( hipStreamSynchronize( (hipStream_t) stream ) );
	return FLANHIP_OK;
	}

static int poll_int_flag( void * user ) { volatile int * c = static_cast<volatile int*>( user ); return c && *c != 0; }

int flanhip_wait_cancellable( void * stream, volatile int * cancel )
	{
	if( int rc = require_device() ) return rc;
	return wait_cancellable( (hipStream_t) stream, cancel ? poll_int_flag : nullptr, const_cast<int*>( cancel ) );
	}

int flanhip_wait_cancellable_fn( void * stream, int ( *poll )( void * ), void * user )
	{
	if( int rc = require_device() ) return rc;
	return wait_cancellable( (hipStream_t) stream, poll, user );
	}

} // extern "C"
