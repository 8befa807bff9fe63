// core.hip -- error state, device plumbing, plan cache, shape helpers of the C ABI (include/flanhip.h).
#include "flanhip_internal.h"
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

static std::mutex g_plan_mutex;   // FFTHelper.cpp:9 serialises plan creation the same way
static std::map<std::tuple<int, int, int>, Plan> g_plans;

int get_plan( int window_size, int dft_size, const Plan ** out )
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	std::lock_guard<std::mutex> lock( g_plan_mutex );
	const auto key = std::make_tuple( device, window_size, dft_size );
	auto it = g_plans.find( key );
	if( it != g_plans.end() ) { *out = &it->second; return FLANHIP_OK; }

	const int C = dft_size / 2;
	std::vector<float> win( window_size );
	for( int i = 0; i < window_size; ++i ) win[i] = hann_host( float( i ) / float( window_size - 1 ) );  // AudioPV.cpp:30-34
	std::vector<cf> tw( C ), tw2( C + 1 );
	const double pi = 3.14159265358979323846;
	for( int k = 0; k < C; ++k ) tw[k] = cf{ float( std::cos( -2.0 * pi * k / C ) ), float( std::sin( -2.0 * pi * k / C ) ) };
	for( int k = 0; k <= C; ++k ) tw2[k] = cf{ float( std::cos( -pi * k / C ) ), float( std::sin( -pi * k / C ) ) };

	Plan plan;
	FLANHIP_CHECK( hipMalloc( &plan.d_window, sizeof( float ) * window_size ) );
	FLANHIP_CHECK( hipMalloc( &plan.d_tw, sizeof( cf ) * C ) );
	FLANHIP_CHECK( hipMalloc( &plan.d_tw2, sizeof( cf ) * ( C + 1 ) ) );
	FLANHIP_CHECK( hipMemcpy( plan.d_window, win.data(), sizeof( float ) * window_size, hipMemcpyHostToDevice ) );
	FLANHIP_CHECK( hipMemcpy( plan.d_tw, tw.data(), sizeof( cf ) * C, hipMemcpyHostToDevice ) );
	FLANHIP_CHECK( hipMemcpy( plan.d_tw2, tw2.data(), sizeof( cf ) * ( C + 1 ), hipMemcpyHostToDevice ) );
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
	auto ins = g_plans.emplace( key, plan );
	*out = &ins.first->second;
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
	std::lock_guard<std::mutex> lock( g_plan_mutex );
	auto it = g_div_plans.find( key );
	if( it != g_div_plans.end() ) { *out = it->second; return FLANHIP_OK; }
	DivPlan d{ c, 1.0f / c, 0 };
	const char * env = std::getenv( "FLANHIP_NO_FAST_DIV" );
	if( !( env && env[0] == '1' ) && c > 1.0e-10f && c < 1.0e10f )
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
	return FLANHIP_OK;
	}

bool force_generic()
	{
	const char * env = std::getenv( "FLANHIP_FORCE_GENERIC" );
	return env && env[0] == '1';
	}

int choose_chain_length( int64_t num_channels, int64_t num_frames, int min_len, int target_chains )
	{
	if( const char * env = std::getenv( "FLANHIP_CHAIN_LEN" ) )
		{
		const int v = std::atoi( env );
		if( v > 0 ) return std::max( v, min_len );
		}
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
	return int( L );
	}

static std::mutex g_ws_mutex;
static std::map<const void*, int> g_ws_producer;
void note_workspace_producer( const void * d_ws, int kind )
	{
	std::lock_guard<std::mutex> lock( g_ws_mutex );
	if( !kind ) { g_ws_producer.erase( d_ws ); return; }
	// addresses the library never sees freed (a caller's own allocator) would pile up: forgetting everything is always safe -- a workspace
	// without a note gets the scan kernel
	if( g_ws_producer.size() >= 1024 ) g_ws_producer.clear();
	g_ws_producer[d_ws] = kind;
	}
int workspace_producer( const void * d_ws )
	{
	std::lock_guard<std::mutex> lock( g_ws_mutex );
	auto it = g_ws_producer.find( d_ws );
	return it == g_ws_producer.end() ? 0 : it->second;
	}

// ---- cancellation inside a launch (defines.h:49-62: the reference polls its flag once per frame, AudioPV.cpp:49,115) -----------------
// Every host thread owns one int of FINE-GRAINED device memory per device (coherent across the chip's XCDs and with the copy engines; page-locked
// coherent host memory if that cannot be had), the CANCEL WORD: the conversion kernels launched by that thread get its address and read it,
// past their caches, when a block starts (the direct-sum kernels also every few batches of frames).  wait_cancellable() waits for a stream
// while it polls the caller's flag; when that rises it sets the word from a side stream, the blocks still to start retire at once (a block of
// the FFT kernels walks at most 512 frames: ~3 ms), and the call returns FLANHIP_ERR_CANCELLED.  The word is the thread's own: a cancelled call
// does not touch what other threads have in flight.
struct CancelWord { int * dev = nullptr; volatile int * host = nullptr; hipStream_t side = nullptr; };   // host != nullptr: the host-memory fallback
struct ThreadCancel { std::map<int, CancelWord> words; };
static thread_local ThreadCancel t_cancel;

static CancelWord * thread_cancel_entry()
	{
	int device = 0;
	if( hipGetDevice( &device ) != hipSuccess ) return nullptr;
	auto it = t_cancel.words.find( device );
	if( it == t_cancel.words.end() )
		{
		CancelWord w;
		void * d = nullptr;
		if( hipExtMallocWithFlags( &d, 256, hipDeviceMallocFinegrained ) == hipSuccess && hipMemset( d, 0, 256 ) == hipSuccess )
			{
			w.dev = static_cast<int*>( d );
			if( hipStreamCreateWithFlags( &w.side, hipStreamNonBlocking ) != hipSuccess ) { (void) hipGetLastError(); (void) hipFree( d ); return nullptr; }
			}
		else
			{
			(void) hipGetLastError();
			void * h = nullptr; void * dp = nullptr;
			if( hipHostMalloc( &h, 256, hipHostMallocCoherent | hipHostMallocMapped ) != hipSuccess ) { (void) hipGetLastError(); return nullptr; }
			std::memset( h, 0, 256 );
			if( hipHostGetDevicePointer( &dp, h, 0 ) != hipSuccess ) { (void) hipGetLastError(); (void) hipHostFree( h ); return nullptr; }
			w.dev = static_cast<int*>( dp ); w.host = static_cast<volatile int*>( h );
			}
		it = t_cancel.words.emplace( device, w ).first;
		}
	return &it->second;
	}

int * thread_cancel_word()
	{
	CancelWord * w = thread_cancel_entry();
	return w ? w->dev : nullptr;
	}

static void set_cancel_word( CancelWord & w, int value, bool wait )
	{
	if( w.host ) { *w.host = value; std::atomic_thread_fence( std::memory_order_seq_cst ); return; }
	(void) hipMemsetAsync( w.dev, value, sizeof( int ), w.side );              // (a byte value: 1 gives 0x01010101 -- non-zero is all that counts)
	if( wait ) (void) hipStreamSynchronize( w.side );
	}

int wait_cancellable( hipStream_t s, int ( *poll )( void * ), void * user )
	{
	CancelWord * w = poll ? thread_cancel_entry() : nullptr;
	if( !w ) { FLANHIP_CHECK( hipStreamSynchronize( s ) ); return ( poll && poll( user ) ) ? FLANHIP_ERR_CANCELLED : FLANHIP_OK; }
	hipEvent_t done;
	FLANHIP_CHECK( hipEventCreateWithFlags( &done, hipEventDisableTiming ) );
	if( hipEventRecord( done, s ) != hipSuccess ) { (void) hipEventDestroy( done ); set_error( "hipEventRecord failed" ); return FLANHIP_ERR_HIP; }
	bool raised = false;
	int spins = 0;
	for( ;; )
		{
		const hipError_t q = hipEventQuery( done );
		if( q == hipSuccess ) break;
		if( q != hipErrorNotReady ) { (void) hipEventDestroy( done ); if( raised ) set_cancel_word( *w, 0, true ); set_error( "hipEventQuery failed" ); return FLANHIP_ERR_HIP; }
		if( !raised && poll( user ) ) { set_cancel_word( *w, 1, false ); raised = true; }
		if( ++spins < 200 ) std::this_thread::yield(); else std::this_thread::sleep_for( std::chrono::microseconds( 50 ) );
		}
	(void) hipEventDestroy( done );
	if( raised ) { set_cancel_word( *w, 0, true ); return FLANHIP_ERR_CANCELLED; }     // (the stream has drained: nobody reads the word now)
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
