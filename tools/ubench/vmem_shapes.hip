// vmem_shapes.hip -- what do the global loads / stores of the analysis kernel cost on their own, by access width?
// One wave per chain of L frames (the kernel's geometry: 8-wave blocks, 2048 chains of 22 frames, 8 channels x 5626 frames):
//   load  A: 16 x 8-byte loads per lane and frame (the 2048 samples of a frame, 512 B contiguous per instruction, frames overlap 75 %)
//   load  B:  8 x 16-byte loads per lane and frame (same bytes)
//   load  C:  4 x 8-byte loads per lane and frame (only the 512 NEW samples of a frame)
//   store A: 17 x 8-byte stores per lane and frame (an MF row of 8200 B, 512 B contiguous per instruction)
//   store B:  8 x 16-byte stores + 1 x 8-byte (same row; rows are only 8-byte aligned)
//   store C: like B with rows padded to 8208 B (16-byte aligned rows) -- not the reference's layout, shows what alignment is worth
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/vmem_shapes.hip -o tools/ubench/vmem_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float v2f __attribute__(( ext_vector_type( 2 ) ));
typedef float v4f __attribute__(( ext_vector_type( 4 ) ));
struct __attribute__(( packed, aligned( 4 ) )) p2 { float x, y; };
struct __attribute__(( packed, aligned( 4 ) )) p4 { float x, y, z, w; };
struct __attribute__(( packed, aligned( 8 ) )) q4 { float x, y, z, w; };

constexpr int CH = 8, F = 5626, L = 22, CPC = ( F + L - 1 ) / L, HOP = 512, BINS = 1025;

template<int MODE>
__global__ __launch_bounds__( 512 ) void k_load( const float * audio, int64_t n, float * sink )
	{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int chain = blockIdx.x * 8 + wave;
	if( chain >= CPC * CH ) return;
	const int ch = chain / CPC, t0 = ( chain % CPC ) * L, t1 = min( t0 + L, F );
	const float * x = audio + int64_t( ch ) * n;
	float acc = 0.0f;
	for( int t = max( t0, 2 ); t < min( t1, F - 3 ); ++t )
		{
		const float * px = x + int64_t( HOP ) * t - 1024;
		if constexpr( MODE == 0 )
			{
			v2f v[16];
			#pragma unroll
			for( int q = 0; q < 16; ++q ) { const p2 a = *reinterpret_cast<const p2*>( px + 2 * lane + 128 * q ); v[q] = v2f{ a.x, a.y }; }
			#pragma unroll
			for( int q = 0; q < 16; ++q ) acc += v[q].x * v[q].y;
			}
		else if constexpr( MODE == 1 )
			{
			v4f v[8];
			#pragma unroll
			for( int q = 0; q < 8; ++q ) { const p4 a = *reinterpret_cast<const p4*>( px + 4 * lane + 256 * q ); v[q] = v4f{ a.x, a.y, a.z, a.w }; }
			#pragma unroll
			for( int q = 0; q < 8; ++q ) acc += v[q].x * v[q].y + v[q].z * v[q].w;
			}
		else
			{
			v2f v[4];
			#pragma unroll
			for( int q = 0; q < 4; ++q ) { const p2 a = *reinterpret_cast<const p2*>( px + 1536 + 2 * lane + 128 * q ); v[q] = v2f{ a.x, a.y }; }
			#pragma unroll
			for( int q = 0; q < 4; ++q ) acc += v[q].x * v[q].y;
			}
		}
	if( acc == 12345.678f ) sink[0] = acc;
	}

template<int MODE>
__global__ __launch_bounds__( 512 ) void k_store( float * pv, float seed )
	{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int chain = blockIdx.x * 8 + wave;
	if( chain >= CPC * CH ) return;
	const int ch = chain / CPC, t0 = ( chain % CPC ) * L, t1 = min( t0 + L, F );
	const int64_t rowlen = MODE == 2 ? 2 * BINS + 2 : 2 * BINS;                // floats per row
	float a = seed + lane;
	for( int t = t0; t < t1; ++t )
		{
		float * row = pv + ( int64_t( ch ) * F + t ) * rowlen;
		a += 1.0f;
		if constexpr( MODE == 0 )
			{
			#pragma unroll
			for( int q = 0; q < 16; ++q ) *reinterpret_cast<v2f*>( row + 2 * ( lane + 64 * q ) ) = v2f{ a, a + q };
			*reinterpret_cast<v2f*>( row + 2048 ) = v2f{ a, a };
			}
		else
			{
			#pragma unroll
			for( int q = 0; q < 8; ++q )
				{
				if constexpr( MODE == 1 ) *reinterpret_cast<q4*>( row + 4 * ( lane + 64 * q ) ) = q4{ a, a + q, a, a - q };
				else *reinterpret_cast<v4f*>( row + 4 * ( lane + 64 * q ) ) = v4f{ a, a + q, a, a - q };
				}
			*reinterpret_cast<v2f*>( row + 2048 ) = v2f{ a, a };
			}
		}
	}

template<class K, class... A> static float time_it( K kern, int blocks, A... args )
	{
	hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
	for( int i = 0; i < 50; ++i ) kern<<<blocks, 512>>>( args... );
	hipEventRecord( e0 );
	for( int i = 0; i < 50; ++i ) kern<<<blocks, 512>>>( args... );
	hipEventRecord( e1 ); hipEventSynchronize( e1 );
	float ms; hipEventElapsedTime( &ms, e0, e1 );
	return ms / 50.0f;
	}

int main()
	{
	const int64_t n = 2880000;
	float * audio, * pv, * sink;
	hipMalloc( &audio, sizeof( float ) * CH * n ); hipMemset( audio, 0, sizeof( float ) * CH * n );
	hipMalloc( &pv, sizeof( float ) * CH * int64_t( F ) * ( 2 * BINS + 2 ) ); hipMalloc( &sink, 64 );
	const int blocks = ( CPC * CH + 7 ) / 8;
	const double mb_pv = double( CH ) * F * BINS * 8 / 1e6, mb_au = double( CH ) * F * 8192 / 1e6;
	printf( "%d blocks of 8 waves, %d chains of %d frames; MF rows %.0f MB, sample reads %.0f MB (L1 level)\n", blocks, CPC * CH, L, mb_pv, mb_au );
	float ms;
	ms = time_it( k_load<0>, blocks, audio, n, sink );  printf( "load  16 x  8 B / lane / frame : %.4f ms  (%.2f TB/s at L1)\n", ms, mb_au / ms / 1e6 );
	ms = time_it( k_load<1>, blocks, audio, n, sink );  printf( "load   8 x 16 B / lane / frame : %.4f ms  (%.2f TB/s at L1)\n", ms, mb_au / ms / 1e6 );
	ms = time_it( k_load<2>, blocks, audio, n, sink );  printf( "load   4 x  8 B (new samples)  : %.4f ms\n", ms );
	ms = time_it( k_store<0>, blocks, pv, 1.0f );      printf( "store 17 x  8 B / lane / frame : %.4f ms  (%.2f TB/s)\n", ms, mb_pv / ms / 1e6 );
	ms = time_it( k_store<1>, blocks, pv, 1.0f );      printf( "store  8 x 16 B + 8 B          : %.4f ms  (%.2f TB/s)\n", ms, mb_pv / ms / 1e6 );
	ms = time_it( k_store<2>, blocks, pv, 1.0f );      printf( "store  8 x 16 B, rows 16-B aligned : %.4f ms  (%.2f TB/s)\n", ms, mb_pv / ms / 1e6 );
	return 0;
	}
