// transfer.hip -- host-side plumbing behind the C ABI: a persistent worker pool, and transfers between ordinary (pageable) host
// memory and the device.  No kernels here.
//
// The reference keeps its buffers in std::vector, so a maintainer who binds flanhip_analyze / flanhip_synthesize straight into
// Flan's classes moves every call's data across PCIe from and to pageable memory.  Measured on the MI355X box
// (tools/ubench/transfer_probe.hip): the runtime's pageable copies run at the link's rate (56 GB/s either way) once the host pages
// exist; what costs is a FRESH destination -- one thread faulting hundreds of MB in, page by page, under the copy (14-26 GB/s).  So a
// download first lets the pool's workers touch the destination's pages together, then hands the copy to the runtime.  (A pipeline of
// 8 MB slabs through page-locked blocks with the pool copying in and out was tried: 52 GB/s down, 30 GB/s up -- no better.)
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>

#include "flanhip_internal.h"

namespace flanhip {

namespace {

int usable_cores()
	{
	if( const char * e = std::getenv( "FLAN_HOST_THREADS" ) ) { const int v = std::atoi( e ); if( v > 0 ) return v; }
	int n = int( std::thread::hardware_concurrency() );
	cpu_set_t set;
	if( sched_getaffinity( 0, sizeof( set ), &set ) == 0 ) n = std::min( n, CPU_COUNT( &set ) );
	std::ifstream quota( "/sys/fs/cgroup/cpu.max" );                                // "<quota> <period>" or "max <period>"
	std::string q; long period = 0;
	if( quota >> q >> period && q != "max" && period > 0 ) n = int( std::min<long>( n, std::max<long>( 1, ( std::atol( q.c_str() ) + period - 1 ) / period ) ) );
	return std::max( 1, std::min( n, 64 ) );
	}

thread_local bool tls_is_worker = false;

struct Pool
	{
	std::vector<std::thread> threads;
	std::mutex job_mutex;                          // one parallel region at a time; a second caller runs its region inline
	std::mutex m;
	std::condition_variable wake, done;
	uint64_t generation = 0;
	void ( *fn )( void *, int ) = nullptr;
	void * ctx = nullptr;
	int n = 0;
	std::atomic<int> next{ 0 };
	int active = 0;

	Pool()
		{
		const int workers = usable_cores() - 1;      // the calling thread works too
		for( int w = 0; w < workers; ++w ) threads.emplace_back( [this]{ worker(); } );
		}
	void drain()
		{
		for( ;; )
			{
			const int i = next.fetch_add( 1, std::memory_order_relaxed );
			if( i >= n ) return;
			fn( ctx, i );
			}
		}
	void worker()
		{
		tls_is_worker = true;
		uint64_t seen = 0;
		std::unique_lock<std::mutex> l( m );
		for( ;; )
			{
			wake.wait( l, [&]{ return generation != seen; } );
			seen = generation;
			l.unlock();
			drain();
			l.lock();
			if( --active == 0 ) done.notify_one();
			}
		}
	};

// never destroyed: the workers sleep on it until the process exits, and objects of the caller's with static lifetime may still use it
Pool & pool() { static Pool * p = new Pool; return *p; }

} // namespace

} // namespace flanhip

using namespace flanhip;

extern "C" {

int flanhip_host_workers( void ) { return int( pool().threads.size() ) + 1; }

int flanhip_parallel_for( int n_tasks, void ( *fn )( void *, int ), void * ctx )
	{
	FLANHIP_REQUIRE( fn, FLANHIP_ERR_INVALID_ARG, "null task" );
	if( n_tasks <= 0 ) return FLANHIP_OK;
	Pool & p = pool();
	std::unique_lock<std::mutex> job( p.job_mutex, std::try_to_lock );
	if( !job || tls_is_worker || p.threads.empty() || n_tasks == 1 ) { for( int i = 0; i < n_tasks; ++i ) fn( ctx, i ); return FLANHIP_OK; }
		{
		std::lock_guard<std::mutex> g( p.m );
		p.fn = fn; p.ctx = ctx; p.n = n_tasks; p.next.store( 0 ); p.active = int( p.threads.size() );
		++p.generation;
		}
	p.wake.notify_all();
	p.drain();
	std::unique_lock<std::mutex> l( p.m );
	p.done.wait( l, [&]{ return p.active == 0; } );
	return FLANHIP_OK;
	}

int flanhip_touch_pages( void * ptr, size_t bytes )
	{
	if( !ptr || bytes < ( size_t( 4 ) << 20 ) ) return FLANHIP_OK;
	struct Job { volatile char * p; size_t bytes; } job{ static_cast<volatile char*>( ptr ), bytes };
	constexpr size_t kRun = size_t( 2 ) << 20;
	return flanhip_parallel_for( int( ( bytes + kRun - 1 ) / kRun ), []( void * ctx, int i )
		{
		const Job & j = *static_cast<const Job*>( ctx );
		const size_t lo = size_t( i ) * kRun, hi = std::min( j.bytes, lo + kRun );
		for( size_t at = lo; at < hi; at += 4096 ) j.p[at] = 0;
		}, &job );
	}

int flanhip_download( void * dst, const void * d_src, size_t bytes )
	{
	if( bytes == 0 ) return FLANHIP_OK;
	FLANHIP_REQUIRE( dst && d_src, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	if( int rc = require_device() ) return rc;
	flanhip_touch_pages( dst, bytes );                                                // every byte of it is about to be overwritten
	FLANHIP_CHECK( hipMemcpy( dst, d_src, bytes, hipMemcpyDeviceToHost ) );
	return FLANHIP_OK;
	}

int flanhip_upload( void * d_dst, const void * src, size_t bytes )
	{
	if( bytes == 0 ) return FLANHIP_OK;
	FLANHIP_REQUIRE( d_dst && src, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	if( int rc = require_device() ) return rc;
	FLANHIP_CHECK( hipMemcpy( d_dst, src, bytes, hipMemcpyHostToDevice ) );
	return FLANHIP_OK;
	}

} // extern "C"
