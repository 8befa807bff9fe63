// simd_census.hip -- how does the dispatcher place the wavefronts of a workgroup on the 4 SIMDs of a CU?
// Every wave records its HW_ID; the host prints, per block size, the histogram of "waves per SIMD" patterns seen per workgroup,
// and times an ILP-4 fma loop (ns per wave-instruction per SIMD) for the same launch shape.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/simd_census.hip -o tools/ubench/simd_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include <algorithm>

__global__ void census( unsigned * ids, float * out, int iters )
	{
	unsigned hw;
	asm volatile( "s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"( hw ) );
	const int wave = threadIdx.x >> 6;
	if( ( threadIdx.x & 63 ) == 0 ) ids[blockIdx.x * ( blockDim.x >> 6 ) + wave] = hw;
	float a = threadIdx.x * 1e-3f + 1.0f, b = a + 1, c = a + 2, d = a + 3;
	for( int it = 0; it < iters; ++it )
		{
		#pragma unroll
		for( int r = 0; r < 32; ++r )
			{
			a = __builtin_fmaf( a, 0.999f, 0.25f ); b = __builtin_fmaf( b, 0.999f, 0.25f );
			c = __builtin_fmaf( c, 0.999f, 0.25f ); d = __builtin_fmaf( d, 0.999f, 0.25f );
			}
		}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
	}

int main()
	{
	unsigned * d_ids; float * d_out;
	hipMalloc( &d_ids, 1 << 20 ); hipMalloc( &d_out, 64 << 20 );
	for( int i = 0; i < 30; ++i ) census<<<256, 1024>>>( d_ids, d_out, 4000 );
	hipDeviceSynchronize();
	const int shapes[][2] = { { 512, 256 }, { 768, 256 }, { 384, 512 }, { 192, 1024 }, { 256, 768 }, { 1024, 256 }, { 640, 256 }, { 896, 256 } };
	for( auto & sh : shapes )
		{
		const int threads = sh[0], blocks = sh[1], wpb = threads / 64, iters = 4000;
		hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
		for( int i = 0; i < 5; ++i ) census<<<blocks, threads>>>( d_ids, d_out, iters );
		hipEventRecord( e0 );
		for( int i = 0; i < 5; ++i ) census<<<blocks, threads>>>( d_ids, d_out, iters );
		hipEventRecord( e1 ); hipEventSynchronize( e1 );
		float ms; hipEventElapsedTime( &ms, e0, e1 );
		std::vector<unsigned> ids( size_t( blocks ) * wpb );
		hipMemcpy( ids.data(), d_ids, ids.size() * 4, hipMemcpyDeviceToHost );
		std::map<std::string, int> pat;
		std::map<unsigned, std::vector<int>> per_cu;                 // key: everything but wave / simd id
		for( int b = 0; b < blocks; ++b )
			{
			int cnt[4] = { 0, 0, 0, 0 };
			for( int w = 0; w < wpb; ++w )
				{
				const unsigned hw = ids[size_t( b ) * wpb + w];
				const int simd = ( hw >> 4 ) & 3;
				++cnt[simd];
				auto & v = per_cu[hw & ~0x3Fu];
				if( v.empty() ) v.assign( 4, 0 );
				++v[simd];
				}
			char s[64]; snprintf( s, sizeof s, "%d%d%d%d", cnt[0], cnt[1], cnt[2], cnt[3] );
			++pat[s];
			}
		std::map<std::string, int> cupat;
		for( auto & kv : per_cu ) { char s[64]; snprintf( s, sizeof s, "%d-%d-%d-%d", kv.second[0], kv.second[1], kv.second[2], kv.second[3] ); ++cupat[s]; }
		const double waves_per_cu = double( blocks ) * wpb / 256.0;
		const double ns = double( ms ) * 1e6 / ( 5.0 * iters * 128.0 * waves_per_cu / 4.0 );
		printf( "block %4d threads x %4d blocks (%.1f waves per SIMD if even): %.2f ns per wave-instruction per SIMD\n   per-workgroup SIMD patterns:", threads, blocks, waves_per_cu / 4, ns );
		for( auto & kv : pat ) printf( " %s x%d", kv.first.c_str(), kv.second );
		printf( "\n   per-CU totals (distinct HW_ID groups %zu):", per_cu.size() );
		int shown = 0;
		for( auto & kv : cupat ) { if( shown++ < 8 ) printf( " %s x%d", kv.first.c_str(), kv.second ); }
		printf( "\n" );
		}
	return 0;
	}
