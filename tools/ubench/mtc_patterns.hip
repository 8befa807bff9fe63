// mtc_patterns.hip -- which thread -> data mapping gets the stretch x 2 of a PV (k_modify_time_chains: read MF[ch][F][1025] and a map[F][1025],
// write MF[ch][2F][1025], every output frame from one pair of input frames) closest to the chip's copy rate?  Memory behaviour only: the
// arithmetic is a stand-in (a weighted sum), the geometry is config 3's (8 ch x 5626 -> 11252 frames, chains of 44 output frames).
//   A  thread = (channel, chain, bin), 256-thread blocks over the linear index, XCD-contiguous block order; 4 input frames per trip (the product kernel)
//   B  block = (channel, chain): 1024 threads own bins 0..1023 and walk the chain's rows together (8200 contiguous bytes per row and block);
//      bin 1024 is walked by a thread per (channel, chain) in a few extra blocks
//   C  like B with 512-thread blocks (two half rows)
//   D  like A with 8 frames per trip;  E like A with normal (not non-temporal) stores
//   Z  the yardstick: a grid-stride copy with the same bytes (16 B per lane)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mtc_patterns.hip -o tools/ubench/mtc_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef float cf __attribute__(( ext_vector_type( 2 ) ));
typedef float v4f __attribute__(( ext_vector_type( 4 ) ));
constexpr int CH = 8, F = 5626, FO = 2 * F, BINS = 1025;
__host__ __device__ constexpr int cpc( int L ) { return ( FO + L - 1 ) / L; }

// MODE: 0 loads + stores, 1 stores only, 2 loads only
template<int B, bool NT, int MODE = 0>
__device__ __forceinline__ void walk( const cf * ip, const float * mp, cf * o, int x_lo, int x_hi, int stride )
	{
	if( MODE == 1 )
		{
		cf v = cf{ float( x_lo ), 1.0f };
		for( int x = x_lo; x < x_hi; ++x ) { if( NT ) __builtin_nontemporal_store( v, o ); else *o = v; o += stride; v.x += 1.0f; }
		return;
		}
	float sink = 0.0f;
	// pair k = ( k-1, k ) feeds output frames 2(k-1)+1 .. 2k  (ceil( 2 (k-1) ) .. ceil( 2 k ) exclusive of the first, as the product's x2 map does)
	int k = x_lo / 2 + 1;
	cf l = ip[int64_t( k - 1 ) * stride];
	int x = x_lo;
	while( x < x_hi )
		{
		cf mf[B]; float tm[B];
		#pragma unroll
		for( int j = 0; j < B; ++j ) { const int64_t kk = int64_t( min( k + j, F - 1 ) ) * stride; mf[j] = ip[kk]; tm[j] = mp[kk]; }
		#pragma unroll
		for( int j = B - 1; j >= 0; --j ) asm volatile( "" : "+v"( tm[j] ), "+v"( mf[j] ) );
		#pragma unroll
		for( int j = 0; j < B; ++j )
			{
			#pragma unroll
			for( int u = 0; u < 2; ++u )
				if( x < x_hi )
					{
					const float w = tm[j] * ( u ? 0.25f : 0.75f );
					const cf v = cf{ l.x * w + mf[j].x, l.y * w + mf[j].y };
					if( MODE == 2 ) sink += v.x + v.y;
					else if( NT ) __builtin_nontemporal_store( v, o ); else *o = v;
					o += stride; ++x;
					}
			l = mf[j];
			}
		k += B;
		}
	if( MODE == 2 && sink == 12345.678f ) *o = cf{ sink, sink };
	}

// ORDER: 0 XCD-contiguous runs of blocks, 1 natural block order, 2 chain-fastest (consecutive blocks = consecutive chains of the same 256 bins)
template<int B, bool NT, int L = 44, int MODE = 0, int ORDER = 0>
__global__ __launch_bounds__( 256 ) void k_A( const cf * in, const float * mod, cf * out )
	{
	constexpr int CPC = cpc( L );
	const int64_t nblocks = gridDim.x, per_xcd = ( nblocks + 7 ) / 8;
	const int64_t vblock = ORDER == 0 ? int64_t( blockIdx.x % 8 ) * per_xcd + blockIdx.x / 8 : int64_t( blockIdx.x );
	int64_t idx = vblock * 256 + threadIdx.x;
	if( ORDER == 2 )
		{
		// pieces of 256 bins (the last piece: 1 bin): block -> ( piece, channel, chain ) with the chain fastest
		constexpr int PIECES = 5;
		const int64_t cc = vblock % ( int64_t( CH ) * CPC ), piece = vblock / ( int64_t( CH ) * CPC );
		const int bin = int( piece ) * 256 + threadIdx.x;
		if( piece >= PIECES || bin >= BINS ) return;
		idx = cc * BINS + bin;
		}
	const int64_t per_channel = int64_t( CPC ) * BINS;
	if( !( int64_t( blockIdx.x / 8 ) < per_xcd && idx < per_channel * CH ) ) return;
	const int channel = int( idx / per_channel ), chain = int( ( idx % per_channel ) / BINS ), bin = int( idx % BINS );
	const int x_lo = chain * L, x_hi = min( x_lo + L, FO );
	walk<B, NT, MODE>( in + int64_t( channel ) * F * BINS + bin, mod + bin, out + ( int64_t( channel ) * FO + x_lo ) * BINS + bin, x_lo, x_hi, BINS );
	}

// two adjacent bins per thread, both columns in step: 16 bytes per lane and store (8-byte aligned), 1 KB per wavefront and instruction
struct __attribute__(( packed, aligned( 8 ) )) q4 { float x, y, z, w; };
template<int B, bool NT, int L = 44>
__global__ __launch_bounds__( 256 ) void k_A2( const cf * in, const float * mod, cf * out )
	{
	constexpr int CPC = cpc( L ), PAIRS = 513;                                        // pairs of bins per row (the last: bin 1024 alone)
	const int64_t nblocks = gridDim.x, per_xcd = ( nblocks + 7 ) / 8;
	const int64_t vblock = int64_t( blockIdx.x % 8 ) * per_xcd + blockIdx.x / 8;
	const int64_t idx = vblock * 256 + threadIdx.x;
	const int64_t per_channel = int64_t( CPC ) * PAIRS;
	if( !( int64_t( blockIdx.x / 8 ) < per_xcd && idx < per_channel * CH ) ) return;
	const int channel = int( idx / per_channel ), chain = int( ( idx % per_channel ) / PAIRS ), pr = int( idx % PAIRS );
	const int x_lo = chain * L, x_hi = min( x_lo + L, FO );
	const bool two = pr < 512;
	const int bin = 2 * pr;
	const cf * ip = in + int64_t( channel ) * F * BINS + bin;
	const float * mp = mod + bin;
	cf * o = out + ( int64_t( channel ) * FO + x_lo ) * BINS + bin;
	int k = x_lo / 2 + 1;
	q4 l = two ? *reinterpret_cast<const q4*>( ip + int64_t( k - 1 ) * BINS ) : q4{ ip[int64_t( k - 1 ) * BINS].x, ip[int64_t( k - 1 ) * BINS].y, 0, 0 };
	int x = x_lo;
	while( x < x_hi )
		{
		q4 mf[B]; cf tm[B];
		#pragma unroll
		for( int j = 0; j < B; ++j )
			{
			const int64_t kk = int64_t( min( k + j, F - 1 ) ) * BINS;
			if( two ) { mf[j] = *reinterpret_cast<const q4*>( ip + kk ); tm[j] = cf{ mp[kk], mp[kk + 1] }; }
			else { const cf a = ip[kk]; mf[j] = q4{ a.x, a.y, 0, 0 }; tm[j] = cf{ mp[kk], 0 }; }
			}
		#pragma unroll
		for( int j = B - 1; j >= 0; --j ) asm volatile( "" : "+v"( tm[j] ), "+v"( mf[j].x ), "+v"( mf[j].y ), "+v"( mf[j].z ), "+v"( mf[j].w ) );
		#pragma unroll
		for( int j = 0; j < B; ++j )
			{
			#pragma unroll
			for( int u = 0; u < 2; ++u )
				if( x < x_hi )
					{
					const float w0 = tm[j].x * ( u ? 0.25f : 0.75f ), w1 = tm[j].y * ( u ? 0.25f : 0.75f );
					const q4 v = q4{ l.x * w0 + mf[j].x, l.y * w0 + mf[j].y, l.z * w1 + mf[j].z, l.w * w1 + mf[j].w };
					if( two ) { if( NT ) __builtin_nontemporal_store( v4f{ v.x, v.y, v.z, v.w }, reinterpret_cast<v4f __attribute__(( aligned( 8 ) ))*>( o ) ); else *reinterpret_cast<q4*>( o ) = v; }
					else *o = cf{ v.x, v.y };
					o += BINS; ++x;
					}
			l = mf[j];
			}
		k += B;
		}
	}

// blocks [0, CH * CPC * (1024 / T)): T threads on a T-bin piece of a chain's rows; the blocks behind them: bin 1024, a thread per (channel, chain)
template<int T, int B, bool NT, int L = 44>
__global__ __launch_bounds__( T ) void k_B( const cf * in, const float * mod, cf * out )
	{
	constexpr int PIECES = 1024 / T, CPC = cpc( L );
	const int main_blocks = CH * CPC * PIECES;
	int channel, chain, bin;
	if( int( blockIdx.x ) < main_blocks )
		{
		const int cc = blockIdx.x / PIECES;
		channel = cc / CPC; chain = cc % CPC; bin = ( blockIdx.x % PIECES ) * T + threadIdx.x;
		}
	else
		{
		const int cc = ( blockIdx.x - main_blocks ) * T + threadIdx.x;
		if( cc >= CH * CPC ) return;
		channel = cc / CPC; chain = cc % CPC; bin = 1024;
		}
	const int x_lo = chain * L, x_hi = min( x_lo + L, FO );
	walk<B, NT>( in + int64_t( channel ) * F * BINS + bin, mod + bin, out + ( int64_t( channel ) * FO + x_lo ) * BINS + bin, x_lo, x_hi, BINS );
	}

__global__ __launch_bounds__( 256 ) void k_Z( const v4f * in, v4f * out, int64_t n_in )      // out[i], out[n + i] from in[i]: three streams
	{
	const int64_t step = int64_t( gridDim.x ) * 256;
	int64_t i = int64_t( blockIdx.x ) * 256 + threadIdx.x;
	for( ; i + 3 * step < n_in; i += 4 * step )
		{
		v4f v[4];
		#pragma unroll
		for( int u = 0; u < 4; ++u ) v[u] = in[i + u * step];
		#pragma unroll
		for( int u = 0; u < 4; ++u ) { out[i + u * step] = v[u]; out[n_in + i + u * step] = v[u] * 0.5f; }
		}
	for( ; i < n_in; i += step ) { const v4f v = in[i]; out[i] = v; out[n_in + i] = v * 0.5f; }
	}

template<class Fn> static float time_ms( Fn fn, int reps = 20 )
	{
	hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
	for( int i = 0; i < 3; ++i ) fn();
	std::vector<float> t;
	for( int r = 0; r < 5; ++r )
		{
		hipEventRecord( e0 );
		for( int i = 0; i < reps; ++i ) fn();
		hipEventRecord( e1 ); hipEventSynchronize( e1 );
		float ms; hipEventElapsedTime( &ms, e0, e1 ); t.push_back( ms / reps );
		}
	std::sort( t.begin(), t.end() );
	return t[2];
	}

int main()
	{
	const size_t n_in = size_t( CH ) * F * BINS, n_out = size_t( CH ) * FO * BINS;
	cf * in, * out; float * mod;
	hipMalloc( &in, n_in * 8 ); hipMalloc( &out, n_out * 8 ); hipMalloc( &mod, size_t( F ) * BINS * 4 );
	hipMemset( in, 0, n_in * 8 ); hipMemset( mod, 0, size_t( F ) * BINS * 4 );
	const double mb = ( n_in * 8 + n_out * 8 + double( F ) * BINS * 4 ) / 1e6;
	// clock warm-up
	for( int i = 0; i < 300; ++i ) hipLaunchKernelGGL( k_Z, dim3( 256 * 16 ), dim3( 256 ), 0, nullptr, (const v4f*) in, (v4f*) out, int64_t( n_in / 2 ) );
	hipDeviceSynchronize();
	auto report = [&]( const char * name, float ms, double bytes_mb ) { printf( "%-56s %.4f ms  %.2f TB/s\n", name, ms, bytes_mb / ms / 1e3 ); };
	const double mb_w = n_out * 8 / 1e6, mb_r = mb - mb_w;
	auto gridA = [&]( int L ) { return unsigned( ( ( int64_t( CH ) * cpc( L ) * BINS + 255 ) / 256 + 7 ) / 8 * 8 ); };
	auto gridA2 = [&]( int L ) { return unsigned( ( ( int64_t( CH ) * cpc( L ) * 513 + 255 ) / 256 + 7 ) / 8 * 8 ); };
	auto gridO2 = [&]( int L ) { return unsigned( CH * cpc( L ) * 5 ); };
	#define RUN( name, bytes, ... ) report( name, time_ms( [&] { hipLaunchKernelGGL( __VA_ARGS__, 0, nullptr, in, mod, out ); } ), bytes )
	report( "Z  grid-stride copy, 16 B per lane, 4 in flight", time_ms( [&] { hipLaunchKernelGGL( k_Z, dim3( 256 * 16 ), dim3( 256 ), 0, nullptr, (const v4f*) in, (v4f*) out, int64_t( n_in / 2 ) ); } ), mb - double( F ) * BINS * 4 / 1e6 );
	RUN( "A  (chain, bin) threads, 256/block, B=4 nt", mb, ( k_A<4, true> ), dim3( gridA( 44 ) ), dim3( 256 ) );
	RUN( "E  ... B=4 plain stores", mb, ( k_A<4, false> ), dim3( gridA( 44 ) ), dim3( 256 ) );
	RUN( "E8 ... B=8 plain stores", mb, ( k_A<8, false> ), dim3( gridA( 44 ) ), dim3( 256 ) );
	RUN( "E, stores only (plain)", mb_w, ( k_A<4, false, 44, 1> ), dim3( gridA( 44 ) ), dim3( 256 ) );
	RUN( "A, stores only (nt)", mb_w, ( k_A<4, true, 44, 1> ), dim3( gridA( 44 ) ), dim3( 256 ) );
	RUN( "E, loads only", mb_r, ( k_A<4, false, 44, 2> ), dim3( gridA( 44 ) ), dim3( 256 ) );
	RUN( "E, natural block order", mb, ( k_A<4, false, 44, 0, 1> ), dim3( gridA( 44 ) ), dim3( 256 ) );
	RUN( "E, chain-fastest block order", mb, ( k_A<4, false, 44, 0, 2> ), dim3( gridO2( 44 ) ), dim3( 256 ) );
	RUN( "E, L = 22", mb, ( k_A<4, false, 22> ), dim3( gridA( 22 ) ), dim3( 256 ) );
	RUN( "E, L = 88", mb, ( k_A<4, false, 88> ), dim3( gridA( 88 ) ), dim3( 256 ) );
	RUN( "E, L = 176", mb, ( k_A<4, false, 176> ), dim3( gridA( 176 ) ), dim3( 256 ) );
	RUN( "E, L = 88, B = 8", mb, ( k_A<8, false, 88> ), dim3( gridA( 88 ) ), dim3( 256 ) );
	RUN( "G  two bins per thread (16 B), B=4 plain", mb, ( k_A2<4, false> ), dim3( gridA2( 44 ) ), dim3( 256 ) );
	RUN( "G  two bins per thread (16 B), B=4 nt", mb, ( k_A2<4, true> ), dim3( gridA2( 44 ) ), dim3( 256 ) );
	RUN( "G  two bins per thread (16 B), B=2 plain", mb, ( k_A2<2, false> ), dim3( gridA2( 44 ) ), dim3( 256 ) );
	RUN( "G  two bins per thread, B=4 plain, L = 88", mb, ( k_A2<4, false, 88> ), dim3( gridA2( 88 ) ), dim3( 256 ) );
	report( "Z  again", time_ms( [&] { hipLaunchKernelGGL( k_Z, dim3( 256 * 16 ), dim3( 256 ), 0, nullptr, (const v4f*) in, (v4f*) out, int64_t( n_in / 2 ) ); } ), mb - double( F ) * BINS * 4 / 1e6 );
	return 0;
	}
