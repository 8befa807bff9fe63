// device_block.h -- RAII handle on a block of HBM obtained through the C ABI (flanhip_malloc / flanhip_free).
#pragma once
#include <atomic>
#include <cstddef>
#include <iostream>
#include <memory>

#include "flanhip.h"

namespace flan { namespace detail {

// host_runtime.cpp: blocks come from, and go back to, a cache of idle HBM blocks (every method synchronises before it returns,
// so a block whose owner dies is idle); flanhip_malloc / flanhip_free only when the cache cannot serve.
void * device_acquire( size_t bytes, size_t * capacity, int * device );   // nullptr on failure (flanhip_last_error() says why)
void device_release( void * ptr, size_t capacity, int device ) noexcept;
void device_cache_flush() noexcept;
// Transfers between ordinary (pageable) host memory and the device (flanhip_download / flanhip_upload).  Synchronous.
bool download_to_host( void * dst, const void * d_src, size_t bytes );
bool upload_from_host( void * d_dst, const void * src, size_t bytes );
void touch_pages( void * p, size_t bytes );                   // first touch of fresh memory on all workers (the kernel zeroes pages per toucher)

struct CopyStreams { void * down = nullptr, * up = nullptr; };  // one per direction, created once; both null when creation failed
CopyStreams copy_streams();                          // give every idle block back to the device

struct DeviceBlock
	{
	void * ptr = nullptr;
	size_t bytes = 0;                                          // what was asked for
	size_t capacity = 0;                                       // what the block holds
	int device = 0;                                            // where it lives
	DeviceBlock() = default;
	DeviceBlock( const DeviceBlock & ) = delete;
	DeviceBlock & operator=( const DeviceBlock & ) = delete;
	~DeviceBlock() { if( ptr ) device_release( ptr, capacity, device ); }

	static std::shared_ptr<DeviceBlock> allocate( size_t bytes )
		{
		auto b = std::make_shared<DeviceBlock>();
		b->ptr = device_acquire( bytes, &b->capacity, &b->device );
		if( !b->ptr )
			{
			std::cerr << "flan: device allocation of " << bytes << " bytes failed: " << flanhip_last_error() << std::endl;
			return nullptr;
			}
		b->bytes = bytes;
		return b;
		}
	};

// the predicate flanhip_wait_cancellable_fn polls: the reference's canceller (defines.h:49-62) is a std::atomic<bool>&
inline int poll_canceller( void * user ) { return static_cast<std::atomic<bool>*>( user )->load() ? 1 : 0; }

inline bool report( int rc, const char * what )
	{
	if( rc == FLANHIP_OK ) return true;
	if( rc != FLANHIP_ERR_CANCELLED )
		std::cerr << "flan: " << what << " failed (" << rc << "): " << flanhip_last_error() << std::endl;
	return false;
	}

} }
