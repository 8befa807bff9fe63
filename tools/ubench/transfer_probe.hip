// transfer_probe.hip -- where the time of a device -> pageable-host transfer goes: the link into page-locked memory, the CPU copy out
// of it (by the pool behind flanhip_parallel_for), the runtime's own pageable copy, and flanhip_download.  hipcc -O2, links libflanhip.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
#include "flanhip.h"

using clk = std::chrono::steady_clock;
static double ms( clk::time_point a, clk::time_point b ) { return std::chrono::duration<double, std::milli>( b - a ).count(); }

struct Job { char * dst; const char * src; size_t bytes, piece; };
static void copy_piece( void * ctx, int i )
	{
	const Job & j = *static_cast<const Job*>( ctx );
	const size_t lo = size_t( i ) * j.piece, n = j.bytes - lo < j.piece ? j.bytes - lo : j.piece;
	std::memcpy( j.dst + lo, j.src + lo, n );
	}

int main()
	{
	const size_t bytes = size_t( 256 ) << 20;
	void * d = nullptr; hipMalloc( &d, bytes ); hipMemset( d, 1, bytes );
	std::vector<char> host( bytes, 0 );
	std::printf( "workers: %d\n", flanhip_host_workers() );
	for( unsigned flags : { unsigned( hipHostMallocDefault ), unsigned( hipHostMallocNonCoherent ), unsigned( hipHostMallocNumaUser ) } )
		{
		void * pinned = nullptr;
		if( hipHostMalloc( &pinned, bytes, flags ) != hipSuccess ) { std::printf( "flags %u: allocation failed\n", flags ); (void) hipGetLastError(); continue; }
		std::memset( pinned, 0, bytes );
		for( int rep = 0; rep < 3; ++rep )
			{
			const auto t0 = clk::now();
			hipMemcpy( pinned, d, bytes, hipMemcpyDeviceToHost );
			const auto t1 = clk::now();
			Job job{ host.data(), static_cast<const char*>( pinned ), bytes, size_t( 256 ) << 10 };
			flanhip_parallel_for( int( bytes / job.piece ), copy_piece, &job );
			const auto t2 = clk::now();
			std::memcpy( host.data(), pinned, bytes );
			const auto t3 = clk::now();
			Job back{ static_cast<char*>( pinned ), host.data(), bytes, size_t( 256 ) << 10 };
			flanhip_parallel_for( int( bytes / back.piece ), copy_piece, &back );
			const auto t4 = clk::now();
			std::printf( "flags %u: link -> pinned %.1f GB/s | pinned -> pageable, pool %.1f GB/s, one thread %.1f GB/s | pageable -> pinned, pool %.1f GB/s\n", flags,
				bytes * 1e-6 / ms( t0, t1 ), bytes * 1e-6 / ms( t1, t2 ), bytes * 1e-6 / ms( t2, t3 ), bytes * 1e-6 / ms( t3, t4 ) );
			}
		hipHostFree( pinned );
		}
	for( int rep = 0; rep < 3; ++rep )
		{
		const auto t0 = clk::now();
		hipMemcpy( host.data(), d, bytes, hipMemcpyDeviceToHost );
		const auto t1 = clk::now();
		flanhip_download( host.data(), d, bytes );
		const auto t2 = clk::now();
		hipMemcpy( d, host.data(), bytes, hipMemcpyHostToDevice );
		const auto t3 = clk::now();
		flanhip_upload( d, host.data(), bytes );
		const auto t4 = clk::now();
		std::printf( "pageable: runtime D2H %.1f GB/s, flanhip_download %.1f GB/s | runtime H2D %.1f GB/s, flanhip_upload %.1f GB/s\n",
			bytes * 1e-6 / ms( t0, t1 ), bytes * 1e-6 / ms( t1, t2 ), bytes * 1e-6 / ms( t2, t3 ), bytes * 1e-6 / ms( t3, t4 ) );
		}
	return 0;
	}
