// host_runtime.cpp -- what sampling a Function on the host (Function.h:141-171 of the reference) needs to keep up with the device:
// a persistent worker pool (the reference gets one from its parallel STL runtime; spawning threads per call costs more than the
// sampling) and a cache of page-locked staging blocks (a grid sampled straight into one uploads at the link's rate; a fresh
// pageable vector pays the page faults, the zero fill and the runtime's staging copy).
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include <sched.h>

#include "device_block.h"
#include "flan/Function.h"
#include "flanhip.h"

namespace flan { namespace detail {

namespace {

int usable_cores()
	{
	if( const char * e = std::getenv( "FLAN_HOST_THREADS" ) ) { const int v = std::atoi( e ); if( v > 0 ) return v; }
	int n = int( std::thread::hardware_concurrency() );
	cpu_set_t set;
	if( sched_getaffinity( 0, sizeof( set ), &set ) == 0 ) n = std::min( n, CPU_COUNT( &set ) );
	std::ifstream quota( "/sys/fs/cgroup/cpu.max" );                                // "<quota> <period>" or "max <period>"
	std::string q; long period = 0;
	if( quota >> q >> period && q != "max" && period > 0 ) n = std::min<long>( n, std::max<long>( 1, ( std::atol( q.c_str() ) + period - 1 ) / period ) );
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
	bool stop = false;
	void ( *fn )( void *, int ) = nullptr;
	void * ctx = nullptr;
	int n = 0;
	std::atomic<int> next{ 0 };
	int active = 0;
	std::exception_ptr error;

	Pool()
		{
		const int workers = usable_cores() - 1;      // the calling thread works too
		for( int w = 0; w < workers; ++w ) threads.emplace_back( [this]{ worker(); } );
		}
	~Pool()
		{
		{ std::lock_guard<std::mutex> g( m ); stop = true; }
		wake.notify_all();
		for( auto & t : threads ) t.join();
		}
	void drain()
		{
		for( ;; )
			{
			const int i = next.fetch_add( 1, std::memory_order_relaxed );
			if( i >= n ) return;
			try { fn( ctx, i ); }
			catch( ... ) { std::lock_guard<std::mutex> g( m ); if( !error ) error = std::current_exception(); }
			}
		}
	void worker()
		{
		tls_is_worker = true;
		uint64_t seen = 0;
		std::unique_lock<std::mutex> l( m );
		for( ;; )
			{
			wake.wait( l, [&]{ return stop || generation != seen; } );
			if( stop ) return;
			seen = generation;
			l.unlock();
			drain();
			l.lock();
			if( --active == 0 ) done.notify_one();
			}
		}
	};

// the three singletons below are never destroyed: objects of the user's with static lifetime may outlive any static of ours,
// and the device runtime may be gone by the time statics die -- the process exit reclaims threads and memory
Pool & pool() { static Pool * p = new Pool; return *p; }

// ---- page-locked staging blocks ----
struct Staging
	{
	struct Block { size_t capacity; bool pinned; };
	std::mutex m;
	std::unordered_map<void*, Block> live;                       // blocks handed out
	std::vector<std::pair<void*, size_t>> spare;                 // pinned blocks waiting for the next grid
	size_t spare_bytes = 0;
	static constexpr size_t kSmall = 256u << 10;                 // below this a grid is not worth a pinned block
	static constexpr size_t kKeep = 1u << 30;                    // at most this much kept for reuse
	};
Staging & staging() { static Staging * s = new Staging; return *s; }

// ---- idle HBM blocks ----
// The methods allocate their results (hundreds of MB for a minute of audio) and free their inputs call after call with the same
// few sizes; the driver's allocate / free pair costs far more than the kernels between them now and then.
struct DeviceCache
	{
	std::mutex m;
	std::vector<std::pair<void*, size_t>> spare;
	size_t spare_bytes = 0;
	size_t keep = size_t( 16 ) << 30;                            // FLAN_DEVICE_CACHE_MB overrides; 0 = no cache
	DeviceCache() { if( const char * e = std::getenv( "FLAN_DEVICE_CACHE_MB" ) ) keep = size_t( std::max( 0L, std::atol( e ) ) ) << 20; }
	static size_t round_up( size_t bytes )
		{
		const size_t grain = bytes >= ( size_t( 1 ) << 20 ) ? size_t( 2 ) << 20 : 4096;
		return ( std::max<size_t>( bytes, 1 ) + grain - 1 ) / grain * grain;
		}
	};
DeviceCache & device_cache() { static DeviceCache * c = new DeviceCache; return *c; }

} // namespace

void * device_acquire( size_t bytes, size_t * capacity )
	{
	DeviceCache & c = device_cache();
	const size_t want = DeviceCache::round_up( bytes );
		{
		std::lock_guard<std::mutex> g( c.m );
		size_t best = c.spare.size();
		for( size_t i = 0; i < c.spare.size(); ++i )               // smallest idle block that fits without wasting more than an eighth
			if( c.spare[i].second >= want && c.spare[i].second - want <= want / 8 && ( best == c.spare.size() || c.spare[i].second < c.spare[best].second ) ) best = i;
		if( best != c.spare.size() )
			{
			void * p = c.spare[best].first;
			*capacity = c.spare[best].second;
			c.spare_bytes -= *capacity;
			c.spare[best] = c.spare.back(); c.spare.pop_back();
			return p;
			}
		}
	void * p = nullptr;
	if( flanhip_malloc( &p, want ) != FLANHIP_OK || !p )
		{
		if( flanhip_device_count() <= 0 ) return nullptr;
		device_cache_flush();                                      // out of memory with idle blocks held back: release them and try once more
		p = nullptr;
		if( flanhip_malloc( &p, want ) != FLANHIP_OK || !p ) return nullptr;
		}
	*capacity = want;
	return p;
	}

void device_release( void * ptr, size_t capacity ) noexcept
	{
	if( !ptr ) return;
	DeviceCache & c = device_cache();
		{
		std::lock_guard<std::mutex> g( c.m );
		if( c.spare_bytes + capacity <= c.keep ) { c.spare.emplace_back( ptr, capacity ); c.spare_bytes += capacity; return; }
		}
	flanhip_free( ptr );
	}

void device_cache_flush() noexcept
	{
	DeviceCache & c = device_cache();
	std::vector<std::pair<void*, size_t>> drop;
		{
		std::lock_guard<std::mutex> g( c.m );
		drop.swap( c.spare );
		c.spare_bytes = 0;
		}
	for( auto & s : drop ) flanhip_free( s.first );
	}

CopyStreams copy_streams()
	{
	static const CopyStreams streams = []
		{
		CopyStreams c;
		if( flanhip_stream_create( &c.down ) != FLANHIP_OK || flanhip_stream_create( &c.up ) != FLANHIP_OK ) c.down = c.up = nullptr;
		return c;
		}();
	return streams;
	}

namespace {
constexpr size_t kSlab = size_t( 8 ) << 20;
constexpr size_t kPiece = size_t( 256 ) << 10;

struct CopyJob { char * dst; const char * src; size_t bytes; };
void parallel_copy( void * dst, const void * src, size_t bytes )
	{
	CopyJob job{ static_cast<char*>( dst ), static_cast<const char*>( src ), bytes };
	pool_run( int( ( bytes + kPiece - 1 ) / kPiece ), []( void * ctx, int i )
		{
		const CopyJob & j = *static_cast<const CopyJob*>( ctx );
		const size_t lo = size_t( i ) * kPiece, n = std::min( kPiece, j.bytes - lo );
		std::memcpy( j.dst + lo, j.src + lo, n );
		}, &job );
	}
}

void touch_pages( void * p, size_t bytes )
	{
	if( bytes < ( size_t( 4 ) << 20 ) ) return;
	struct Job { volatile char * p; size_t bytes; } job{ static_cast<volatile char*>( p ), bytes };
	constexpr size_t kRun = size_t( 2 ) << 20;
	pool_run( int( ( bytes + kRun - 1 ) / kRun ), []( void * ctx, int i )
		{
		const Job & j = *static_cast<const Job*>( ctx );
		const size_t lo = size_t( i ) * kRun, hi = std::min( j.bytes, lo + kRun );
		for( size_t at = lo; at < hi; at += 4096 ) j.p[at] = 0;
		}, &job );
	}

bool download_to_host( void * dst, const void * d_src, size_t bytes )
	{
	if( bytes == 0 ) return true;
	const CopyStreams streams = copy_streams();
	if( bytes < 2 * kSlab || !streams.down )
		return flanhip_memcpy_d2h( dst, d_src, bytes, nullptr ) == FLANHIP_OK && flanhip_stream_synchronize( nullptr ) == FLANHIP_OK;
	if( flanhip_stream_synchronize( nullptr ) != FLANHIP_OK ) return false;           // whatever produced the data has finished
	char * stage[2] = { static_cast<char*>( staging_acquire( kSlab ) ), static_cast<char*>( staging_acquire( kSlab ) ) };
	const size_t slabs = ( bytes + kSlab - 1 ) / kSlab;
	auto fetch = [&]( size_t k )
		{
		const size_t lo = k * kSlab;
		return flanhip_memcpy_d2h( stage[k & 1], static_cast<const char*>( d_src ) + lo, std::min( kSlab, bytes - lo ), streams.down ) == FLANHIP_OK;
		};
	bool ok = fetch( 0 );
	for( size_t k = 0; k < slabs && ok; ++k )
		{
		ok = flanhip_stream_synchronize( streams.down ) == FLANHIP_OK;               // slab k is in its block
		if( ok && k + 1 < slabs ) ok = fetch( k + 1 );                                // the other block: its last reader finished an iteration ago
		if( ok ) parallel_copy( static_cast<char*>( dst ) + k * kSlab, stage[k & 1], std::min( kSlab, bytes - k * kSlab ) );
		}
	flanhip_stream_synchronize( streams.down );
	staging_release( stage[0] ); staging_release( stage[1] );
	return ok;
	}

bool upload_from_host( void * d_dst, const void * src, size_t bytes )
	{
	if( bytes == 0 ) return true;
	const CopyStreams streams = copy_streams();
	if( bytes < 2 * kSlab || !streams.up )
		return flanhip_memcpy_h2d( d_dst, src, bytes, nullptr ) == FLANHIP_OK && flanhip_stream_synchronize( nullptr ) == FLANHIP_OK;
	char * stage[2] = { static_cast<char*>( staging_acquire( kSlab ) ), static_cast<char*>( staging_acquire( kSlab ) ) };
	const size_t slabs = ( bytes + kSlab - 1 ) / kSlab;
	bool ok = true;
	for( size_t k = 0; k < slabs && ok; ++k )
		{
		const size_t lo = k * kSlab, n = std::min( kSlab, bytes - lo );
		parallel_copy( stage[k & 1], static_cast<const char*>( src ) + lo, n );         // while slab k-1 is on the link
		if( k >= 1 ) ok = flanhip_stream_synchronize( streams.up ) == FLANHIP_OK;      // slab k-1 has left the other block (free for k+1)
		if( ok ) ok = flanhip_memcpy_h2d( static_cast<char*>( d_dst ) + lo, stage[k & 1], n, streams.up ) == FLANHIP_OK;
		}
	ok = flanhip_stream_synchronize( streams.up ) == FLANHIP_OK && ok;
	staging_release( stage[0] ); staging_release( stage[1] );
	return ok;
	}

int host_workers() { return int( pool().threads.size() ) + 1; }

void pool_run( int n_tasks, void ( *fn )( void *, int ), void * ctx )
	{
	if( n_tasks <= 0 ) return;
	Pool & p = pool();
	std::unique_lock<std::mutex> job( p.job_mutex, std::try_to_lock );
	if( !job || tls_is_worker || p.threads.empty() || n_tasks == 1 ) { for( int i = 0; i < n_tasks; ++i ) fn( ctx, i ); return; }
		{
		std::lock_guard<std::mutex> g( p.m );
		p.fn = fn; p.ctx = ctx; p.n = n_tasks; p.next.store( 0 ); p.active = int( p.threads.size() ); p.error = nullptr;
		++p.generation;
		}
	p.wake.notify_all();
	p.drain();
	std::unique_lock<std::mutex> l( p.m );
	p.done.wait( l, [&]{ return p.active == 0; } );
	if( p.error ) { auto e = p.error; p.error = nullptr; l.unlock(); std::rethrow_exception( e ); }
	}

void * staging_acquire( size_t bytes )
	{
	Staging & s = staging();
	if( bytes >= Staging::kSmall )
		{
		size_t capacity = Staging::kSmall;
		while( capacity < bytes ) capacity <<= 1;
			{
			std::lock_guard<std::mutex> g( s.m );
			for( size_t i = 0; i < s.spare.size(); ++i )
				if( s.spare[i].second == capacity )
					{
					void * p = s.spare[i].first;
					s.spare_bytes -= capacity;
					s.spare[i] = s.spare.back(); s.spare.pop_back();
					s.live[p] = { capacity, true };
					return p;
					}
			}
		void * p = nullptr;
		if( flanhip_host_malloc( &p, capacity ) == FLANHIP_OK && p )
			{
			std::lock_guard<std::mutex> g( s.m );
			s.live[p] = { capacity, true };
			return p;
			}
		}
	void * p = std::malloc( bytes ? bytes : 1 );                  // small, or no device: ordinary memory
	if( !p ) throw std::bad_alloc();
	return p;
	}

void staging_release( void * p ) noexcept
	{
	if( !p ) return;
	Staging & s = staging();
	bool pinned = false; size_t capacity = 0; bool keep = false;
		{
		std::lock_guard<std::mutex> g( s.m );
		auto it = s.live.find( p );
		if( it != s.live.end() )
			{
			pinned = true; capacity = it->second.capacity;
			s.live.erase( it );
			if( s.spare_bytes + capacity <= Staging::kKeep ) { s.spare.emplace_back( p, capacity ); s.spare_bytes += capacity; keep = true; }
			}
		}
	if( !pinned ) std::free( p );
	else if( !keep ) flanhip_host_free( p );
	}

} }
