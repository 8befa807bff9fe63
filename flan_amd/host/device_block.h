// device_block.h -- RAII handle on a block of HBM obtained through the C ABI (flanhip_malloc / flanhip_free).
#pragma once
#include <cstddef>
#include <iostream>
#include <memory>

#include "flanhip.h"

namespace flan { namespace detail {

struct DeviceBlock
	{
	void * ptr = nullptr;
	size_t bytes = 0;
	DeviceBlock() = default;
	DeviceBlock( const DeviceBlock & ) = delete;
	DeviceBlock & operator=( const DeviceBlock & ) = delete;
	~DeviceBlock() { if( ptr ) flanhip_free( ptr ); }

	static std::shared_ptr<DeviceBlock> allocate( size_t bytes )
		{
		auto b = std::make_shared<DeviceBlock>();
		if( flanhip_malloc( &b->ptr, bytes ) != FLANHIP_OK )
			{
			std::cerr << "flan: device allocation of " << bytes << " bytes failed: " << flanhip_last_error() << std::endl;
			return nullptr;
			}
		b->bytes = bytes;
		return b;
		}
	};

inline bool report( int rc, const char * what )
	{
	if( rc == FLANHIP_OK ) return true;
	if( rc != FLANHIP_ERR_CANCELLED )
		std::cerr << "flan: " << what << " failed (" << rc << "): " << flanhip_last_error() << std::endl;
	return false;
	}

} }
