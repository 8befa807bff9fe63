// collective.hip -- the one collective of the path: reassembling the channel shards of the output.
//
// Channels shard across GPUs with no exchange during compute; the north star's output step is ONE in-place all-gather of
// float[channels_per_gpu][frames] per rank into float[all channels][frames] -- the layout is channel-major, so the gathered
// buffer IS the final AudioBuffer.  This file binds RCCL (the collective library over xGMI) at run time: libflanhip.so has no
// link-time dependency on it, single-GPU users never load it, and a process that already holds an RCCL (PyTorch bundles one)
// shares it instead of loading a second copy.
#include "flanhip_internal.h"
#include <dlfcn.h>
#include <cstring>

namespace flanhip {

struct RcclUniqueId { char internal[FLANHIP_COMM_ID_BYTES]; };                  // ncclUniqueId, rccl.h:40-43

struct Rccl
	{
	int ( *GetUniqueId )( RcclUniqueId * ) = nullptr;                              // ncclGetUniqueId
	int ( *CommInitRank )( void **, int, RcclUniqueId, int ) = nullptr;            // ncclCommInitRank, rccl.h:220
	int ( *CommDestroy )( void * ) = nullptr;                                      // ncclCommDestroy
	int ( *AllGather )( const void *, void *, size_t, int, void *, hipStream_t ) = nullptr;   // ncclAllGather, rccl.h:678
	const char * ( *GetErrorString )( int ) = nullptr;
	bool ok = false;
	};

static const Rccl & rccl()
	{
	static Rccl r = []
		{
		Rccl t;
		void * h = nullptr;
		for( const char * name : { "librccl.so", "librccl.so.1" } )                // an RCCL this process already loaded, if any
			if( ( h = dlopen( name, RTLD_NOW | RTLD_NOLOAD ) ) ) break;
		if( !h )
			for( const char * name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" } )
				if( ( h = dlopen( name, RTLD_NOW | RTLD_LOCAL ) ) ) break;
		if( !h ) return t;
		t.GetUniqueId    = reinterpret_cast<decltype( t.GetUniqueId )>( dlsym( h, "ncclGetUniqueId" ) );
		t.CommInitRank   = reinterpret_cast<decltype( t.CommInitRank )>( dlsym( h, "ncclCommInitRank" ) );
		t.CommDestroy    = reinterpret_cast<decltype( t.CommDestroy )>( dlsym( h, "ncclCommDestroy" ) );
		t.AllGather      = reinterpret_cast<decltype( t.AllGather )>( dlsym( h, "ncclAllGather" ) );
		t.GetErrorString = reinterpret_cast<decltype( t.GetErrorString )>( dlsym( h, "ncclGetErrorString" ) );
		t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.AllGather;
		return t;
		}();
	return r;
	}

static int rccl_check( int rc, const char * what )
	{
	if( rc == 0 ) return FLANHIP_OK;
	const Rccl & r = rccl();
	set_error( "%s failed: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString( rc ) : "?" );
	return FLANHIP_ERR_HIP;
	}

#define FLANHIP_NEED_RCCL() \
	do { if( !rccl().ok ) { set_error( "%s: librccl.so could not be loaded", __func__ ); return FLANHIP_ERR_UNSUPPORTED; } } while( 0 )

} // namespace flanhip

using namespace flanhip;

extern "C" {

int flanhip_comm_unique_id( char * id_out )
	{
	FLANHIP_REQUIRE( id_out, FLANHIP_ERR_INVALID_ARG, "null id" );
	FLANHIP_NEED_RCCL();
	RcclUniqueId id;
	if( int rc = rccl_check( rccl().GetUniqueId( &id ), "ncclGetUniqueId" ) ) return rc;
	std::memcpy( id_out, id.internal, FLANHIP_COMM_ID_BYTES );
	return FLANHIP_OK;
	}

int flanhip_comm_init( const char * id, int world_size, int rank, void ** comm_out )
	{
	FLANHIP_REQUIRE( id && comm_out && world_size >= 1 && rank >= 0 && rank < world_size, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	if( int rc = require_device() ) return rc;
	FLANHIP_NEED_RCCL();
	RcclUniqueId uid;
	std::memcpy( uid.internal, id, FLANHIP_COMM_ID_BYTES );
	void * comm = nullptr;
	if( int rc = rccl_check( rccl().CommInitRank( &comm, world_size, uid, rank ), "ncclCommInitRank" ) ) return rc;
	*comm_out = comm;
	return FLANHIP_OK;
	}

int flanhip_comm_destroy( void * comm )
	{
	if( !comm ) return FLANHIP_OK;
	FLANHIP_NEED_RCCL();
	return rccl_check( rccl().CommDestroy( comm ), "ncclCommDestroy" );
	}

int flanhip_allgather_audio( void * comm, float * d_all, int64_t count_per_rank, int rank, void * stream )
	{
	FLANHIP_REQUIRE( comm && d_all && count_per_rank > 0 && rank >= 0, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	FLANHIP_NEED_RCCL();
	const int nccl_float = 7;                                                      // ncclFloat32, rccl.h:466
	// in place: this rank's shard already sits at its final position (the documented in-place form of ncclAllGather)
	return rccl_check( rccl().AllGather( d_all + size_t( rank ) * size_t( count_per_rank ), d_all, size_t( count_per_rank ), nccl_float, comm,
		(hipStream_t) stream ), "ncclAllGather" );
	}

} // extern "C"
