// scan_pipe.hip -- where does k_stretch_map's time go?  The product's column_scan_piped (processors_common.h) on config 3's map (5626 frames x
// 1025 bins) as it ships and with one side of the pipeline switched off: the scanning wavefront idle (movers alone), the movers touching no
// memory (the scan and the barriers alone), both off (the barriers alone).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iflan_amd/csrc -Iinclude tools/ubench/scan_pipe.hip -o tools/ubench/scan_pipe
#include "processors_common.h"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace flanhip;

template<int TF, int TBc, int THREADS, int PROBE>
__global__ __launch_bounds__( THREADS ) void k_map( float * factor, int64_t F, int bins, float sr, float hop, float * sink )
	{
	__shared__ __attribute__(( aligned( 16 ) )) float lds[3 * column_scan_lds_floats( TF, TBc, 1 )];
	const int strip = xcd_contiguous_strip( blockIdx.x, ( bins + TBc - 1 ) / TBc );
	if( strip < 0 ) return;
	const int bin = min( strip * TBc + int( threadIdx.x % TBc ), bins - 1 );
	auto at = [&]( int64_t f ) -> float * { return factor + unsigned( int( f ) * bins + bin ); };
	float run = -0.0f, mx = -INFINITY;
	column_scan_piped<TF, TBc, THREADS, PROBE>( lds, F, at,
		[&]( float v ) { run = v + run; return run; },
		[&]( int64_t, float v ) { const float t = frame_to_time( v, sr, hop ); mx = fmaxf( mx, t ); return t; } );
	if( mx == 12345.678f ) *sink = mx + run;
	}

template<int TF, int TBc, int THREADS, int PROBE>
static void run( const char * name, float * d, float * sink, int64_t F, int bins )
	{
	hipEvent_t e0, e1;
	hipEventCreate( &e0 ); hipEventCreate( &e1 );
	std::vector<float> ms;
	const dim3 grid( xcd_grid( ( bins + TBc - 1 ) / TBc ) );
	for( int r = 0; r < 12; ++r )
		{
		hipMemsetAsync( d, 0, sizeof( float ) * F * bins, 0 );
		hipEventRecord( e0, 0 );
		hipLaunchKernelGGL( ( k_map<TF, TBc, THREADS, PROBE> ), grid, dim3( THREADS ), 0, 0, d, F, bins, 48000.0f, 512.0f, sink );
		hipEventRecord( e1, 0 );
		hipEventSynchronize( e1 );
		float t; hipEventElapsedTime( &t, e0, e1 ); ms.push_back( t );
		}
	std::sort( ms.begin(), ms.end() );
	printf( "%-58s tile %3d x %2d, %4d threads: median %7.2f us  min %7.2f us\n", name, TF, TBc, THREADS, ms[ms.size() / 2] * 1e3, ms[0] * 1e3 );
	}

// one wavefront alone on its SIMD: cycles per step of NCHAIN interleaved dependent chains of v_add_f32 (s_memtime around 64 x 64 steps)
template<int NCHAIN, bool WITH_LDS>
__global__ void k_chain( float * out, long long * cycles, float seed )
	{
	__shared__ float buf[64 * 68];
	float r[NCHAIN];
	for( int c = 0; c < NCHAIN; ++c ) r[c] = seed + c + threadIdx.x;
	for( int i = threadIdx.x; i < 64 * 68; i += 64 ) buf[i] = seed;
	__syncthreads();
	const long long t0 = __builtin_readcyclecounter();
	#pragma unroll 1
	for( int it = 0; it < 64; ++it )
		{
		float x[16];
		if( WITH_LDS ) { for( int j = 0; j < 16; ++j ) x[j] = buf[threadIdx.x * 68 + j]; }
		else { for( int j = 0; j < 16; ++j ) x[j] = seed * j; }
		#pragma unroll
		for( int j = 0; j < 64; ++j )
			#pragma unroll
			for( int c = 0; c < NCHAIN; ++c ) { r[c] = x[( j + c ) & 15] + r[c]; asm volatile( "" : "+v"( r[c] ) ); }
		if( WITH_LDS ) { for( int j = 0; j < 16; ++j ) buf[threadIdx.x * 68 + j] = r[j % NCHAIN]; }
		}
	const long long t1 = __builtin_readcyclecounter();
	float s = 0; for( int c = 0; c < NCHAIN; ++c ) s += r[c];
	out[threadIdx.x] = s;
	if( threadIdx.x == 0 ) *cycles = t1 - t0;
	}
template<int NCHAIN, bool WITH_LDS>
static void chain( float * d, long long * dc )
	{
	long long c = 0;
	for( int r = 0; r < 3; ++r ) { hipLaunchKernelGGL( ( k_chain<NCHAIN, WITH_LDS> ), dim3( 1 ), dim3( 64 ), 0, 0, d, dc, 1.0f ); hipDeviceSynchronize(); }
	hipMemcpy( &c, dc, 8, hipMemcpyDeviceToHost );
	printf( "one wavefront, %d interleaved dependent v_add_f32 chain(s)%s: %.2f counter ticks per step (%.2f per instruction)\n", NCHAIN, WITH_LDS ? " + 16 LDS reads / writes per 64 steps" : "", double( c ) / ( 64.0 * 64.0 ), double( c ) / ( 64.0 * 64.0 * NCHAIN ) );
	}

int main()
	{
	{
	float * dd; long long * dc; hipMalloc( &dd, 256 ); hipMalloc( &dc, 8 );
	chain<1, false>( dd, dc ); chain<2, false>( dd, dc ); chain<4, false>( dd, dc ); chain<1, true>( dd, dc );
	}
	const int64_t F = 5626; const int bins = 1025;
	float * d, * sink;
	hipMalloc( &d, sizeof( float ) * F * bins ); hipMalloc( &sink, 4 );
	for( int rep = 0; rep < 2; ++rep )
		{
		run<224, 16, 512, 0>( "as shipped", d, sink, F, bins );
		run<224, 16, 512, 1>( "scanning wavefront idle (movers + barriers)", d, sink, F, bins );
		run<224, 16, 512, 2>( "movers touch no memory (scan + LDS traffic + barriers)", d, sink, F, bins );
		run<224, 16, 512, 3>( "neither (LDS traffic + barriers)", d, sink, F, bins );
		run<224, 8, 512, 0>( "as shipped, 8 columns per block", d, sink, F, bins );
		run<224, 8, 512, 2>( "movers touch no memory, 8 columns", d, sink, F, bins );
		run<224, 32, 960, 0>( "as shipped, 32 columns per block", d, sink, F, bins );
		run<224, 32, 960, 2>( "movers touch no memory, 32 columns", d, sink, F, bins );
		}
	return 0;
	}
