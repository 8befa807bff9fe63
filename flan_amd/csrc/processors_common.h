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

// The same scan as a three-stage pipeline over tiles, for kernels whose time is the sum of column_scan's three phases rather than any one of
// them (k_stretch_map: a running sum down 5 626 frames with 16 columns per block -- the scan itself is ~2 us per tile, moving the tile in and
// out ~2.5 us each, one after the other).  Wave 0 only scans; the other waves of the block (THREADS - 64 threads) only move.  While tile k is scanned in buffer k % 3, the movers write
// tile k - 1 back from ( k - 1 ) % 3, put the prefetched tile k + 1 into ( k + 1 ) % 3 and request tile k + 2 from memory: one barrier per
// tile, and a tile's three phases overlap with its neighbours'.  lds: 3 * column_scan_lds_floats( TFr, TBc, max( NIN, NOUT ) ) floats.
template<int TFr, int TBc, int NIN, int NOUT, bool REVERSE, int THREADS, class Load, class Step, class Store>
__device__ __forceinline__ void column_scan_piped( float * lds, int64_t F, Load load, Step step, Store store )
	{
	constexpr int NT = NIN > NOUT ? NIN : NOUT, MOVERS = THREADS - 64, NY = MOVERS / TBc, RP = TFr / NY, CH = 32;
	static_assert( TFr % CH == 0 && MOVERS % TBc == 0 && TFr % NY == 0 && TBc <= 64 && 64 % TBc == 0, "tile shape" );
	constexpr int TILE = NT * TBc * ( TFr + 4 );
	auto T = [&]( int buf, int a, int r, int c ) -> float & { return lds[buf * TILE + ( a * TBc + c ) * ( TFr + 4 ) + r]; };
	const bool mover = threadIdx.x >= 64;
	const int tx = threadIdx.x % TBc, ty = mover ? ( int( threadIdx.x ) - 64 ) / TBc : 0;
	const int64_t tiles = ( F + TFr - 1 ) / TFr;
	auto tile_of = [&]( int64_t k ) { return REVERSE ? tiles - 1 - k : k; };
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
	auto regs_to_lds = [&]( int buf )
		{
		#pragma unroll
		for( int a = 0; a < NIN; ++a )
			#pragma unroll
			for( int i = 0; i < RP; ++i ) T( buf, a, ty + NY * i, tx ) = pre[a][i];
		};
	auto write_back = [&]( int buf, int64_t tile_i )
		{
		#pragma unroll
		for( int i = 0; i < RP; ++i )
			{
			const int64_t f = tile_i * TFr + ty + NY * i;
			if( f < F )
				{
				float v[NOUT];
				#pragma unroll
				for( int a = 0; a < NOUT; ++a ) v[a] = T( buf, a, ty + NY * i, tx );
				store( f, v );
				}
			}
		};
	auto scan = [&]( int buf, int64_t tile_i )
		{
		const int64_t fbase = tile_i * TFr;
		#pragma unroll 1
		for( int c0 = 0; c0 < TFr; c0 += CH )
			{
			const int base = REVERSE ? TFr - CH - c0 : c0;
			float col[NT][CH];
			#pragma unroll
			for( int a = 0; a < NIN; ++a )
				#pragma unroll
				for( int j = 0; j < CH; ++j ) col[a][j] = T( buf, a, base + j, tx );
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
				for( int j = 0; j < CH; ++j ) T( buf, a, base + j, tx ) = col[a][j];
			}
		};
	if( tiles <= 0 ) return;
	if( mover ) { load_regs( tile_of( 0 ) ); regs_to_lds( 0 ); if( tiles > 1 ) load_regs( tile_of( 1 ) ); }
	__syncthreads();
	for( int64_t k = 0; k < tiles; ++k )
		{
		const int b = int( k % 3 );
		if( mover )
			{
			if( k >= 1 ) write_back( ( b + 2 ) % 3, tile_of( k - 1 ) );
			if( k + 1 < tiles )
				{
				regs_to_lds( ( b + 1 ) % 3 );
				if( k + 2 < tiles ) load_regs( tile_of( k + 2 ) );
				}
			}
		else if( threadIdx.x < TBc ) scan( b, tile_of( k ) );
		__syncthreads();
		}
	if( mover ) write_back( int( ( tiles - 1 ) % 3 ), tile_of( tiles - 1 ) );
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
