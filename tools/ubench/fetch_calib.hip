// fetch_calib.hip -- what do FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for a KNOWN byte count in the access shapes of the
// PV kernels?  MI355X_MICROARCH.md (HBM): 16 B/lane streaming reads count half on gfx950, other widths are uncalibrated -> calibrate.
// Every kernel streams the same 1 GiB (past the 256 MiB Infinity Cache) exactly once:
//   read_b4 / read_b8 / read_b16      : 4 / 8 / 16 bytes per lane, consecutive lanes consecutive addresses, plain loads
//   read_b8_nt                        : 8 bytes per lane, non-temporal loads (k_synthesize_v2's row loads)
//   read_rows                         : a wave walks 8200-byte rows with 17 x 8-byte loads per lane (lanes 0 of the 17th only), 8-byte
//                                       aligned rows -- k_synthesize_v2's exact row pattern
//   write_b8 / write_b8_nt / write_b16: 8 / 16 bytes per lane stores (k_analyze_v2 stores MF rows with 8-byte non-temporal stores)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/ubench/fetch_calib      (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

namespace flanhip_calib {

typedef float v2f __attribute__(( ext_vector_type( 2 ) ));
typedef float v4f __attribute__(( ext_vector_type( 4 ) ));

template<typename T, bool NT>
__global__ __launch_bounds__( 256 ) void k_read( const T * __restrict__ p, int64_t count, float * sink )
	{
	float acc = 0.0f;
	for( int64_t i = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x; i < count; i += int64_t( gridDim.x ) * blockDim.x )
		{
		T v;
		if constexpr( NT ) v = __builtin_nontemporal_load( p + i ); else v = p[i];
		if constexpr( sizeof( T ) == 4 ) acc += v;
		else if constexpr( sizeof( T ) == 8 ) acc += v.x + v.y;
		else acc += v.x + v.y + v.z + v.w;
		}
	if( acc == 12345.678f ) sink[0] = acc;
	}

__global__ __launch_bounds__( 512 ) void k_read_rows( const float * __restrict__ p, int64_t rows, float * sink )
	{
	const int lane = threadIdx.x & 63;
	const int64_t wave = int64_t( blockIdx.x ) * 8 + ( threadIdx.x >> 6 ), waves = int64_t( gridDim.x ) * 8;
	float acc = 0.0f;
	for( int64_t r = wave; r < rows; r += waves )
		{
		const v2f * row = reinterpret_cast<const v2f*>( p + r * 2050 );
		#pragma unroll
		for( int q = 0; q < 16; ++q ) { const v2f v = __builtin_nontemporal_load( row + lane + 64 * q ); acc += v.x + v.y; }
		if( lane == 0 ) { const v2f v = __builtin_nontemporal_load( row + 1024 ); acc += v.x + v.y; }
		}
	if( acc == 12345.678f ) sink[0] = acc;
	}

template<typename T, bool NT>
__global__ __launch_bounds__( 256 ) void k_write( T * __restrict__ p, int64_t count )
	{
	for( int64_t i = int64_t( blockIdx.x ) * blockDim.x + threadIdx.x; i < count; i += int64_t( gridDim.x ) * blockDim.x )
		{
		T v;
		if constexpr( sizeof( T ) == 8 ) v = T{ float( i ), 1.0f }; else v = T{ float( i ), 1.0f, 2.0f, 3.0f };
		if constexpr( NT ) __builtin_nontemporal_store( v, p + i ); else p[i] = v;
		}
	}

} // namespace flanhip_calib

using namespace flanhip_calib;

int main()
	{
	const int64_t bytes = int64_t( 1 ) << 30;
	float * buf = nullptr; float * sink = nullptr;
	if( hipMalloc( &buf, bytes + 8200 ) != hipSuccess || hipMalloc( &sink, 64 ) != hipSuccess ) { printf( "hipMalloc failed\n" ); return 1; }
	(void) hipMemset( buf, 0, bytes + 8200 );
	const int grid = 256 * 16;
	for( int rep = 0; rep < 3; ++rep )
		{
		hipLaunchKernelGGL( ( k_read<float, false> ), dim3( grid ), dim3( 256 ), 0, 0, buf, bytes / 4, sink );
		hipLaunchKernelGGL( ( k_read<v2f, false> ), dim3( grid ), dim3( 256 ), 0, 0, reinterpret_cast<const v2f*>( buf ), bytes / 8, sink );
		hipLaunchKernelGGL( ( k_read<v4f, false> ), dim3( grid ), dim3( 256 ), 0, 0, reinterpret_cast<const v4f*>( buf ), bytes / 16, sink );
		hipLaunchKernelGGL( ( k_read<v2f, true> ), dim3( grid ), dim3( 256 ), 0, 0, reinterpret_cast<const v2f*>( buf ), bytes / 8, sink );
		hipLaunchKernelGGL( k_read_rows, dim3( 2048 ), dim3( 512 ), 0, 0, buf, bytes / 8200, sink );
		hipLaunchKernelGGL( ( k_write<v2f, false> ), dim3( grid ), dim3( 256 ), 0, 0, reinterpret_cast<v2f*>( buf ), bytes / 8 );
		hipLaunchKernelGGL( ( k_write<v2f, true> ), dim3( grid ), dim3( 256 ), 0, 0, reinterpret_cast<v2f*>( buf ), bytes / 8 );
		hipLaunchKernelGGL( ( k_write<v4f, false> ), dim3( grid ), dim3( 256 ), 0, 0, reinterpret_cast<v4f*>( buf ), bytes / 16 );
		}
	if( hipDeviceSynchronize() != hipSuccess ) { printf( "kernel failed\n" ); return 1; }
	printf( "every kernel moved %lld bytes (read_rows: %lld)\n", (long long) bytes, (long long) ( bytes / 8200 * 8200 ) );
	return 0;
	}
