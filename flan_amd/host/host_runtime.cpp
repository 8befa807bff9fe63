// host_runtime.cpp -- what sampling a Function on the host (Function.h:141-171 of the reference) needs to keep up with the device:
// the persistent worker pool behind the C ABI (flanhip_parallel_for; the reference gets one from its parallel STL runtime, and
// spawning threads per call costs more than the sampling), a cache of page-locked staging blocks (a grid sampled straight into one
// uploads at the link's rate; a fresh pageable vector pays the page faults, the zero fill and the runtime's staging copy) and a
// cache of idle HBM blocks.
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
	struct Idle { void * ptr; size_t capacity; int device; };
	std::vector<Idle> spare;                                     // a block goes back only to the device it came from
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

void * device_acquire( size_t bytes, size_t * capacity, int * device )
	{
	DeviceCache & c = device_cache();
	const size_t want = DeviceCache::round_up( bytes );
	int current = 0;
	if( flanhip_get_device( &current ) != FLANHIP_OK ) return nullptr;                // no device: flanhip_last_error() says so
	*device = current;
		{
		std::lock_guard<std::mutex> g( c.m );
		size_t best = c.spare.size();
		for( size_t i = 0; i < c.spare.size(); ++i )               // smallest idle block of this device that fits without wasting more than an eighth
			if( c.spare[i].device == current && c.spare[i].capacity >= want && c.spare[i].capacity - want <= want / 8
				&& ( best == c.spare.size() || c.spare[i].capacity < c.spare[best].capacity ) ) best = i;
		if( best != c.spare.size() )
			{
			void * p = c.spare[best].ptr;
			*capacity = c.spare[best].capacity;
			c.spare_bytes -= *capacity;
			c.spare[best] = c.spare.back(); c.spare.pop_back();
			return p;
			}
		}
	void * p = nullptr;
	if( flanhip_malloc( &p, want ) != FLANHIP_OK || !p )
		{
		device_cache_flush();                                      // out of memory with idle blocks held back: release them and try once more
		p = nullptr;
		if( flanhip_malloc( &p, want ) != FLANHIP_OK || !p ) return nullptr;
		}
	*capacity = want;
	return p;
	}

void device_release( void * ptr, size_t capacity, int device ) noexcept
	{
	if( !ptr ) return;
	DeviceCache & c = device_cache();
		{
		std::lock_guard<std::mutex> g( c.m );
		if( c.spare_bytes + capacity <= c.keep ) { c.spare.push_back( DeviceCache::Idle{ ptr, capacity, device } ); c.spare_bytes += capacity; return; }
		}
	flanhip_free( ptr );
	}

void device_cache_flush() noexcept
	{
	DeviceCache & c = device_cache();
	std::vector<DeviceCache::Idle> drop;
		{
		std::lock_guard<std::mutex> g( c.m );
		drop.swap( c.spare );
		c.spare_bytes = 0;
		}
	for( auto & s : drop ) flanhip_free( s.ptr );
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

void touch_pages( void * p, size_t bytes ) { flanhip_touch_pages( p, bytes ); }
bool download_to_host( void * dst, const void * d_src, size_t bytes ) { return flanhip_download( dst, d_src, bytes ) == FLANHIP_OK; }
bool upload_from_host( void * d_dst, const void * src, size_t bytes ) { return flanhip_upload( d_dst, src, bytes ) == FLANHIP_OK; }

int host_workers() { return flanhip_host_workers(); }

// the pool lives behind the C ABI (flan_amd/csrc/transfer.hip) and runs plain functions; what a callable throws is carried out of
// the region here and rethrown to the caller of the PV method
void pool_run( int n_tasks, void ( *fn )( void *, int ), void * ctx )
	{
	struct Region { void ( *fn )( void *, int ); void * ctx; std::mutex m; std::exception_ptr error; } region{ fn, ctx, {}, nullptr };
	flanhip_parallel_for( n_tasks, []( void * r, int i )
		{
		Region & reg = *static_cast<Region*>( r );
		try { reg.fn( reg.ctx, i ); }
		catch( ... ) { std::lock_guard<std::mutex> g( reg.m ); if( !reg.error ) reg.error = std::current_exception(); }
		}, &region );
	if( region.error ) std::rethrow_exception( region.error );
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
