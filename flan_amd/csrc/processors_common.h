// processors_common.h -- small pieces shared by the frame-processor translation units (processors.hip, processors_ext.hip).
#pragma once
#include "flanhip_internal.h"
#include "fft_device.h"

namespace flanhip {

struct MFd { float m, f; };

// PVBuffer.cpp:428-431, :433-436, :438-441, :443-446
__device__ __forceinline__ float time_to_frame( float t, float sr, float hop ) { return t * sr / hop; }
__device__ __forceinline__ float frame_to_time( float f, float sr, float hop ) { return f / ( sr / hop ); }
__device__ __forceinline__ float frequency_to_bin( float f, float sr, float dft ) { return f / ( sr / dft ); }
__device__ __forceinline__ float bin_to_frequency( float b, float sr, float dft ) { return b * sr / dft; }

// float -> Bin / Frame.  The reference's plain conversion is undefined outside the int range (x86 yields INT_MIN there and
// for NaN); here it saturates and NaN becomes INT_MIN: every range check downstream rejects all of those values alike.
__device__ __forceinline__ int to_int_sat( float v )
	{
	return ( v == v ) ? int( v ) : INT_MIN;                                        // v_cvt_i32_f32 saturates
	}

// std::clamp( v, 0.0f, 1.0f ): NaN passes through
__device__ __forceinline__ float clamp01( float v ) { return v < 0.0f ? 0.0f : ( 1.0f < v ? 1.0f : v ); }

// An Interpolator built from a user's callable (Utility/Interpolator.h): the host samples it at i / FLANHIP_INTERP_TABLE_INTERVALS, i = 0 ..
// INTERVALS, and at NaN (flanhip_interp_table_create); kinds FLANHIP_INTERP_TABLE_FIRST + slot read the table with linear interpolation
// between neighbouring samples, the argument clamped to [0, 1] (every call site's argument is a position inside a pair: PVModify.cpp:232,
// :344, :491).  The pointers sit in a __device__ array: one copy per translation unit (no relocatable device code), set through
// set_interp_lut_here() from that unit.
constexpr int kInterpLutSlots = 32;
static __device__ const float * g_interp_lut[kInterpLutSlots];
static inline int set_interp_lut_here( int slot, const float * d_table )
	{
	FLANHIP_CHECK( hipMemcpyToSymbol( HIP_SYMBOL( g_interp_lut ), &d_table, sizeof( d_table ), sizeof( d_table ) * size_t( slot ) ) );
	return FLANHIP_OK;
	}
__device__ __forceinline__ float interpolate_table( int slot, float x )
	{
	const float * t = g_interp_lut[slot];
	if( !( x == x ) ) return t[FLANHIP_INTERP_TABLE_INTERVALS + 1];
	const float pos = ( x < 0.0f ? 0.0f : ( 1.0f < x ? 1.0f : x ) ) * float( FLANHIP_INTERP_TABLE_INTERVALS );
	const int i = min( int( pos ), FLANHIP_INTERP_TABLE_INTERVALS - 1 );
	const float a = t[i], b = t[i + 1];
	return __builtin_fmaf( pos - float( i ), b - a, a );
	}

// Utility/Interpolator.cpp:14-101, numbered as include/flanhip.h numbers them (FLANHIP_INTERP_*)
__device__ __forceinline__ float interpolate( int kind, float x )
	{
	if( kind >= FLANHIP_INTERP_TABLE_FIRST ) return interpolate_table( kind - FLANHIP_INTERP_TABLE_FIRST, x );
	switch( kind )
		{
		case FLANHIP_INTERP_MIDPOINT:     return 0.5f;
		case FLANHIP_INTERP_NEAREST:      return roundf( x );
		case FLANHIP_INTERP_FLOOR:        return 0.0f;
		case FLANHIP_INTERP_CEIL:         return 1.0f;
		case FLANHIP_INTERP_SMOOTHSTEP:   return x * x * ( 3.0f - 2.0f * x );
		case FLANHIP_INTERP_SMOOTHERSTEP: return x * x * x * ( x * ( x * 6.0f - 15.0f ) + 10.0f );
		case FLANHIP_INTERP_SQRT:         return sqrtf( x );                                 // correctly rounded (hipcc default)
		case FLANHIP_INTERP_SINE:         return ( 1.0f - cosf( 3.14159274101257324f * x ) ) / 2.0f;   // cosf: within 1-2 ulp of libm's
		default:                          return x;                                                    // linear
		}
	}

// The "placement with conflicts" rule shared by PV::shape with shift alignment (PV.cpp:438-448) and PV::time_extrapolate
// (PVModify.cpp:658-664): candidates are visited in ascending source-bin order and one replaces the occupant of its target
// bin only if its magnitude is STRICTLY greater, the row starting from { 0, 0 }.  The survivor of a target bin is therefore
// the candidate with the greatest magnitude (> 0, not NaN), the lowest source bin among equals.  A wavefront resolves one
// row through LDS: key = ( magnitude bits << 32 | ~source bin ), ds_max_u64 per candidate, then the winners are written.
__device__ __forceinline__ void placement_offer( unsigned long long * keys, int target, float m, int source_bin )
	{
	if( m > 0.0f )                                                                 // beats the initial 0; false for NaN
		atomicMax( &keys[target], ( (unsigned long long) __float_as_uint( m ) << 32 ) | (unsigned long long) ( 0xFFFFFFFFu - unsigned( source_bin ) ) );
	}
__device__ __forceinline__ int placement_winner( unsigned long long key ) { return int( 0xFFFFFFFFu - unsigned( key & 0xFFFFFFFFull ) ); }

// ---------------------------------------------------------------------------------------------------------------------
// Column scans.  Several processors carry a state down the frames of every bin column in a fixed order (a running fp32 sum,
// a selection accumulator, a decaying maximum): sequential per column by definition, and a PV has only ~1000 columns.
// A block of 256 threads owns TBc (16 / 32 / 64) adjacent columns and walks them in tiles of TFr frames (few columns per
// block = many blocks, deep tiles = many bytes in flight per block: these kernels are bound by latency x concurrency):
//   * all 256 threads move the tile between HBM and LDS in coalesced rows, the NEXT tile is already in flight (registers)
//     while the current one is scanned;
//   * TBc threads of wave 0 each scan one column, CH rows at a time: CH batched LDS reads into registers, the
//     recurrence on registers, CH batched writes -- no LDS round trip inside the dependent chain.
// load( frame, v[NIN] ) / store( frame, v[NOUT] ) are called with frame < F for the thread's own column; step( frame, v )
// replaces v[0..NIN) by v[0..NOUT) and is called in scan order (frames past F included: harmless, never stored).
// lds: max(NIN,NOUT) * TBc * (TFr+4) floats, 16-byte aligned (column_scan_lds_floats).  Column of a thread: blockIdx.x * TBc + threadIdx.x % TBc.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TB = 64;

// Workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8) and each XCD has its own L2.  Column strips that are
// neighbours in memory share cache lines (a 16-bin strip is 64 B of every 128-B line), so neighbours should meet in ONE L2:
// XCD x gets the contiguous run of strips [ x * ceil(n/8), ... ).  Returns the strip of this block, or -1 for the few blocks past
// the end when n is not a multiple of 8 (the grid is rounded up to a multiple of 8).
__device__ __forceinline__ int xcd_contiguous_strip( int block, int strips )
	{
	const int per_xcd = ( strips + 7 ) / 8;
	const int strip = ( block % 8 ) * per_xcd + block / 8;
	return ( block / 8 < per_xcd && strip < strips ) ? strip : -1;
	}
inline unsigned xcd_grid( int strips ) { return unsigned( ( ( strips + 7 ) / 8 ) * 8 ); }

__host__ __device__ constexpr int column_scan_lds_floats( int TFr, int TBc, int arrays ) { return arrays * TBc * ( TFr + 4 ); }

template<int TFr, int TBc, int NIN, int NOUT, bool REVERSE, class Load, class Step, class Store>
__device__ __forceinline__ void column_scan( float * lds, int64_t F, Load load, Step step, Store store )
	{
	constexpr int NT = NIN > NOUT ? NIN : NOUT, NY = 256 / TBc, RP = TFr / NY, CH = 32;
	static_assert( TFr % CH == 0 && TFr % NY == 0 && TBc <= 64, "tile shape" );
	// column-major tile: the CH rows a scanning thread reads or writes at a time are contiguous (ds_read_b128 / ds_write_b128)
	auto T = [&]( int a, int r, int c ) -> float & { return lds[( a * TBc + c ) * ( TFr + 4 ) + r]; };
	const int tx = threadIdx.x % TBc, ty = threadIdx.x / TBc;
	const int64_t tiles = ( F + TFr - 1 ) / TFr;
	float pre[NIN][RP];
	auto load_regs = [&]( int64_t tile_i )
		{
		#pragma unroll
		for( int i = 0; i < RP; ++i )
			{
			const int64_t f = tile_i * TFr + ty + NY * i;
			float v[NIN];
			#pragma unroll
			for( int a = 0; a < NIN; ++a ) v[a] = 0.0f;
			if( f < F ) load( f, v );
			#pragma unroll
			for( int a = 0; a < NIN; ++a ) pre[a][i] = v[a];
			}
		};
	load_regs( REVERSE ? tiles - 1 : 0 );
	for( int64_t k = 0; k < tiles; ++k )
		{
		const int64_t tile_i = REVERSE ? tiles - 1 - k : k, fbase = tile_i * TFr;
		#pragma unroll
		for( int a = 0; a < NIN; ++a )
			#pragma unroll
			for( int i = 0; i < RP; ++i ) T( a, ty + NY * i, tx ) = pre[a][i];
		__syncthreads();
		if( k + 1 < tiles ) load_regs( REVERSE ? tile_i - 1 : tile_i + 1 );
		if( threadIdx.x < TBc )
			{
			#pragma unroll 1
			for( int c0 = 0; c0 < TFr; c0 += CH )
				{
				const int base = REVERSE ? TFr - CH - c0 : c0;
				float col[NT][CH];
				#pragma unroll
				for( int a = 0; a < NIN; ++a )
					#pragma unroll
					for( int j = 0; j < CH; ++j ) col[a][j] = T( a, base + j, tx );
				#pragma unroll
				for( int jj = 0; jj < CH; ++jj )
					{
					const int j = REVERSE ? CH - 1 - jj : jj;
					float v[NT];
					#pragma unroll
					for( int a = 0; a < NIN; ++a ) v[a] = col[a][j];
					step( fbase + base + j, v );
					#pragma unroll
					for( int a = 0; a < NOUT; ++a ) col[a][j] = v[a];
					}
				#pragma unroll
				for( int a = 0; a < NOUT; ++a )
					#pragma unroll
					for( int j = 0; j < CH; ++j ) T( a, base + j, tx ) = col[a][j];
				}
			}
		__syncthreads();
		#pragma unroll
		for( int i = 0; i < RP; ++i )
			{
			const int64_t f = fbase + ty + NY * i;
			if( f < F )
				{
				float v[NOUT];
				#pragma unroll
				for( int a = 0; a < NOUT; ++a ) v[a] = T( a, ty + NY * i, tx );
				store( f, v );
				}
			}
		__syncthreads();
		}
	}

// A running fp32 sum down the frames of a grid, column by column and in frame order (sequential: the reference's roundings), as a three-stage
// pipeline over tiles of TFr frames x TBc columns (k_stretch_map: 5 626 frames, 16 columns per block).  Wave 0 only scans -- its time is the
// dependent chain of additions, ~6 cycles a frame; the other waves of the block (THREADS - 64 threads) only move: while tile k is scanned in LDS
// buffer k % 3 they write tile k - 1 back from ( k - 1 ) % 3 and put tile k + 1 into ( k + 1 ) % 3.  The movers' rows are requested THREE tiles
// ahead into three register sets, by hand-written loads and stores with hand-counted waits: a tile's 8 loads are followed in the memory queue by
// 16 younger loads and 16 younger stores, loads and stores retire through ONE in-order counter (vmcnt), so `s_waitcnt vmcnt(32)` is "this
// tile's rows are here" while everything younger stays in flight.  (The compiler's own counting gave vmcnt(0) at every merge of the tail arms: a
// memory round trip per tile, 1.9 us x 26 tiles = the whole kernel.)  Every mover issues the same instructions whatever its rows: rows past F
// read row F - 1, columns are clamped by the caller's addr() (they then store what their neighbour stores); only the LAST tile, whose rows may
// end early, is written back by ordinary predicated stores.  lds: 3 * TBc * ( TFr + 4 ) floats.
// addr( f ): this thread's element of row f; step( v ): the running sum; post( f, v ): the value stored for row f.
__device__ __forceinline__ void asm_load( float & dst, const float * p ) { asm volatile( "global_load_dword %0, %1, off" : "=&v"( dst ) : "v"( p ) : "memory" ); }
__device__ __forceinline__ void asm_store( float * p, float v ) { asm volatile( "global_store_dword %0, %1, off" : : "v"( p ), "v"( v ) : "memory" ); }
template<int N> __device__ __forceinline__ void asm_wait_vm() { asm volatile( "s_waitcnt vmcnt(%0)" : : "n"( N ) : "memory" ); }
__device__ __forceinline__ void lds_barrier()
	{
	__builtin_amdgcn_fence( __ATOMIC_RELEASE, "workgroup", "local" );               // (LDS only: nothing here waits for the movers' memory queue)
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "workgroup", "local" );
	}
// PROBE (tools/ubench/scan_pipe.hip only): 1 = the scanning wavefront does nothing, 2 = the movers touch no memory, 3 = neither
template<int TFr, int TBc, int THREADS, int PROBE = 0, class Addr, class Step, class Post>
__device__ __forceinline__ void column_scan_piped( float * lds, int64_t F, Addr addr, Step step, Post post )
	{
	constexpr int MOVERS = THREADS - 64, NY = MOVERS / TBc, RP = TFr / NY, CH = 32, D = 3;
	static_assert( TFr % CH == 0 && MOVERS % TBc == 0 && TFr % NY == 0 && TBc <= 64 && 64 % TBc == 0 && ( D - 1 ) * 2 * RP <= 63, "tile shape" );
	constexpr int TILE = TBc * ( TFr + 4 );
	auto T = [&]( int buf, int r, int c ) -> float & { return lds[buf * TILE + c * ( TFr + 4 ) + r]; };
	const bool mover = threadIdx.x >= 64;
	const int tx = threadIdx.x % TBc, ty = mover ? ( int( threadIdx.x ) - 64 ) / TBc : 0;
	const int64_t tiles = ( F + TFr - 1 ) / TFr;
	if( tiles <= 0 ) return;
	float pre[D][RP];
	auto request = [&]( int64_t tile_i, float ( &dst )[RP] )                        // always RP loads
		{
		const int64_t f0 = min( tile_i, tiles - 1 ) * TFr + ty;
		#pragma unroll
		for( int i = 0; i < RP; ++i ) { if( PROBE & 2 ) dst[i] = 1.0f; else asm_load( dst[i], addr( min( f0 + NY * i, F - 1 ) ) ); }
		};
	auto to_lds = [&]( int buf, float ( &src )[RP] )
		{
		#pragma unroll
		for( int i = 0; i < RP; ++i ) asm volatile( "" : "+v"( src[i] ) );          // (read below the wait just issued)
		#pragma unroll
		for( int i = 0; i < RP; ++i ) T( buf, ty + NY * i, tx ) = src[i];
		};
	auto write_back_full = [&]( int buf, int64_t tile_i )                           // always RP stores
		{
		const int64_t f0 = tile_i * TFr + ty;
		float v[RP];
		#pragma unroll
		for( int i = 0; i < RP; ++i ) v[i] = T( buf, ty + NY * i, tx );
		#pragma unroll
		for( int i = 0; i < RP; ++i ) { const float t = post( f0 + NY * i, v[i] ); if( !( PROBE & 2 ) ) asm_store( addr( f0 + NY * i ), t ); }
		};
	// the scanning wavefront's tile, the whole tile unrolled.  What it costs (tools/ubench/scan_pipe.hip, profiles/r04_scan_pipe_ubench.txt): one
	// wavefront issues a dependent v_add_f32 every 8.3 cycles and the LDS traffic -- a ds_read_b128 and a ds_write_b128 per four frames -- costs
	// ~11 cycles per instruction whatever the schedule (hand-placed waits and reads a chunk ahead: the same time): ~14 cycles a frame, 33 of the
	// kernel's 42 us.
	auto scan = [&]( int buf )
		{
		constexpr int NCH = TFr / CH;
		float col[2][CH];
		#pragma unroll
		for( int j = 0; j < CH; ++j ) col[0][j] = T( buf, j, tx );
		#pragma unroll
		for( int c = 0; c < NCH; ++c )
			{
			float ( &cur )[CH] = col[c & 1];
			if( c + 1 < NCH )
				{
				#pragma unroll
				for( int j = 0; j < CH; ++j ) col[( c + 1 ) & 1][j] = T( buf, ( c + 1 ) * CH + j, tx );
				}
			#pragma unroll
			for( int j = 0; j < CH; ++j ) cur[j] = step( cur[j] );
			#pragma unroll
			for( int j = 0; j < CH; ++j ) T( buf, c * CH + j, tx ) = cur[j];
			}
		};
	if( mover )
		{
		#pragma unroll
		for( int d = 0; d < D; ++d ) request( d, pre[d] );
		asm_wait_vm<( D - 1 ) * RP>();
		to_lds( 0, pre[0] );
		request( D, pre[0] );
		}
	lds_barrier();
	for( int64_t k0 = 0; k0 < tiles; k0 += D )
		{
		#pragma unroll
		for( int u = 0; u < D; ++u )                                                // (D == 3 == the LDS buffers: buffer and register set are both static)
			{
			const int64_t k = k0 + u;
			if( k >= tiles ) break;
			if( mover )
				{
				if( k + 1 < tiles )
					{
					// tile k + 1's rows: behind them in the queue the loads of tiles k + 2, k + 3 and -- from the third iteration on -- the stores of
					// the two iterations before this one
					if( k >= D ) asm_wait_vm<( D - 1 ) * 2 * RP>(); else asm_wait_vm<( D - 1 ) * RP>();
					to_lds( ( u + 1 ) % 3, pre[( u + 1 ) % D] );
					request( k + 1 + D, pre[( u + 1 ) % D] );
					}
				if( k >= 1 ) write_back_full( ( u + 2 ) % 3, k - 1 );
				}
			else if( threadIdx.x < TBc && !( PROBE & 1 ) ) scan( u );
			lds_barrier();
			}
		}
	if( mover )
		{
		asm_wait_vm<0>();                                                           // (requests past the last tile are still in flight)
		const int buf = int( ( tiles - 1 ) % 3 );
		const int64_t f0 = ( tiles - 1 ) * TFr + ty;
		#pragma unroll
		for( int i = 0; i < RP; ++i )
			{
			const int64_t f = f0 + NY * i;
			if( f < F ) *addr( f ) = post( f, T( buf, ty + NY * i, tx ) );
			}
		}
	}

struct DevBuf
	{
	void * p = nullptr;
	~DevBuf() { if( p ) (void) hipFree( p ); }
	int alloc( size_t bytes ) { FLANHIP_CHECK( hipMalloc( &p, bytes ? bytes : 1 ) ); return FLANHIP_OK; }
	};

// Entry points that need a transient workspace take it from the stream's memory pool (hipMallocAsync).  By default the pool hands
// its memory back to the driver at every synchronisation, so each call would pay for mapping its workspace again (milliseconds for
// hundreds of MB); tell the device's pool once to keep what it has been given.
inline void retain_pool_memory()
	{
	static bool done[64] = {};
	int device = 0;
	if( hipGetDevice( &device ) != hipSuccess || device < 0 || device >= 64 || done[device] ) return;
	hipMemPool_t pool = nullptr;
	if( hipDeviceGetDefaultMemPool( &pool, device ) == hipSuccess && pool )
		{
		uint64_t keep = UINT64_MAX;
		(void) hipMemPoolSetAttribute( pool, hipMemPoolAttrReleaseThreshold, &keep );
		}
	(void) hipGetLastError();
	done[device] = true;
	}

inline int check_pv_args( const void * a, const void * b, int64_t ch, int64_t F, int bins, float sr )
	{
	FLANHIP_REQUIRE( a && b, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	FLANHIP_REQUIRE( ch > 0 && F > 0 && bins >= 2 && sr > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad sizes" );
	return require_device();
	}

} // namespace flanhip
