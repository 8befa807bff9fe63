// processors_arrange.hip -- the PV methods that select, rearrange and re-place frames and bins, behind the C ABI:
// get_frame (PV/PV.cpp:24-39), select (:92-127), freeze (:129-198), cut_frames (:643-668), join (:698-720),
// add_octaves / add_harmonics (:362-419); and, further down, each with its own notes: smear_time (PV/PVModify.cpp:513-605), modify
// (:15-193, the general warp) and stretch_spline (:387-443, with the vendored spline/spline.h).
//
// The first group, all but its last, are gathers: every output MF is one input MF (or a blend of two), so they run at the rate HBM moves 8 B in and
// 8 B out per MF.  A block owns one output row (channel, frame) and walks its bins: no index division per element, rows coalesced.
// add_octaves / add_harmonics scatter every bin to the bins of its overtones with a "strictly louder replaces" rule in a fixed
// visiting order; a block resolves one row through ds_max_u64 keys in LDS ( magnitude bits << 32 | ~visiting index ), like the
// placement rule of shape / time_extrapolate (processors_common.h).
#include "processors_common.h"
#include <algorithm>
#include <vector>

namespace flanhip {

// PV.cpp:24-39 with getBinInterpolated( Channel, float frame, Bin ), :62-73.  grid = ( ceil(bins/256), channels )
__global__ __launch_bounds__( 256 ) void k_get_frame( const MFd * in, int64_t F, int bins, int64_t lo, int64_t hi, float mix, MFd * out )
	{
	const int bin = blockIdx.x * 256 + threadIdx.x;
	if( bin >= bins ) return;
	const int64_t channel = blockIdx.y;
	const MFd l = in[( channel * F + lo ) * bins + bin], h = in[( channel * F + hi ) * bins + bin];
	out[channel * bins + bin] = MFd{ ( 1.0f - mix ) * l.m + mix * h.m, ( 1.0f - mix ) * l.f + mix * h.f };   // :69-72
	}

// out[c][o][:] = in[c][src(o)][:], zero where src(o) is not a frame of the input; src(o) = src[o], or start + o when src is null.
// freeze's copy loops (PV.cpp:176-195) and cut_frames (:660-665).  One block per output row.
__global__ __launch_bounds__( 256 ) void k_select_frames( const MFd * in, int64_t F, int bins, const int * src, int64_t start, int64_t Fo, MFd * out )
	{
	const int64_t row = blockIdx.x;                                                   // channel * Fo + o
	const int64_t o = row % Fo, channel = row / Fo;
	const int64_t s = src ? int64_t( src[o] ) : start + o;
	MFd * op = out + row * bins;
	if( s < 0 || s >= F ) { for( int b = threadIdx.x; b < bins; b += 256 ) op[b] = MFd{ 0.0f, 0.0f }; return; }
	const MFd * ip = in + ( channel * F + s ) * bins;
	for( int b = threadIdx.x; b < bins; b += 256 ) op[b] = ip[b];
	}

// one input of join (PV.cpp:708-716): its rows to out rows [out_start, out_start + in_F), the channels and bins both have
__global__ __launch_bounds__( 256 ) void k_place_frames( const MFd * in, int64_t in_F, int in_bins, MFd * out, int64_t out_F, int out_bins, int64_t out_start, int copy_bins )
	{
	const int64_t row = blockIdx.x;                                                   // channel * in_F + frame
	const int64_t fr = row % in_F, channel = row / in_F;
	const MFd * ip = in + row * in_bins;
	MFd * op = out + ( channel * out_F + out_start + fr ) * out_bins;
	for( int b = threadIdx.x; b < copy_bins; b += 256 ) op[b] = ip[b];
	}

// PV.cpp:106-123.  selector: TF[Fo][bins] as float2 ( t, f ).  One block per output row.
__global__ __launch_bounds__( 256 ) void k_select( const MFd * in, int64_t F, int bins, float sr, float hop, float dft, const float2 * selector, int64_t Fo, MFd * out )
	{
	const int64_t row = blockIdx.x;
	const int64_t fr = row % Fo, channel = row / Fo;
	const float2 * sp = selector + fr * bins;
	MFd * op = out + row * bins;
	for( int b = threadIdx.x; b < bins; b += 256 )
		{
		const float2 s = sp[b];                                                       // :110
		const int sf = to_int_sat( time_to_frame( s.x, sr, hop ) );                   // :111
		const int sb = to_int_sat( frequency_to_bin( s.y, sr, dft ) );                // :112
		MFd m{ 0.0f, 0.0f };
		if( !( sf < 0 || F - 1 <= sf || sb < 0 || bins - 1 <= sb ) )                  // :114-116
			{
			m = in[( channel * F + sf ) * bins + sb];
			if( s.y > 1.0f ) m.f *= bin_to_frequency( float( b ), sr, dft ) / s.y;    // :119-120
			}
		op[b] = m;
		}
	}

// PV.cpp:381-404 for one (channel, frame) row per block.  MODE 0: octaves, f * pow( 2, h ) in double, rounded once (:411);
// MODE 1: harmonics, f * ( h + 1 ) in fp32 (:417), h the 1-based harmonic number.  Visiting order of the reference: source bins
// ascending, harmonics ascending inside; a candidate replaces the occupant only if strictly louder, the row starting from zeros.
template<int MODE>
__device__ __forceinline__ float harmonic_frequency( float f, int hh )
	{
	return MODE == 0 ? float( ldexp( double( f ), hh ) ) : f * float( hh + 1 );
	}

constexpr int kWideBins = 64;              // the lowest bins have up to `bins` overtones each: a wavefront shares one such bin's overtones

// LDS: keys[bins]; when `staged`, also the input row and the row of the series (every offer reads both: from LDS the walk over a
// bin's overtones costs LDS latency per step instead of an L2 round trip).
template<int MODE>
__global__ __launch_bounds__( 256 ) void k_harmonic_scale( const MFd * in, int64_t F, int bins, float sr, float dft, const float * series, int H, int staged, MFd * out )
	{
	extern __shared__ unsigned long long keys[];                                      // [bins] (+ MFd[bins] + float[H])
	const int64_t row = blockIdx.x;
	const int64_t fr = row % F;
	const MFd * ip = in + row * bins;
	const float * sp = series + fr * H;
	for( int b = threadIdx.x; b < bins; b += 256 ) keys[b] = 0ull;
	if( staged )
		{
		MFd * lrow = reinterpret_cast<MFd*>( keys + bins );
		float * lser = reinterpret_cast<float*>( lrow + bins );
		for( int b = threadIdx.x; b < bins; b += 256 ) lrow[b] = ip[b];
		for( int h = threadIdx.x; h < H; h += 256 ) lser[h] = sp[h];
		ip = lrow; sp = lser;
		}
	__syncthreads();
	auto offer = [&]( int b, const MFd source, int h ) -> bool                        // false: this and every later overtone is past the last bin
		{
		const float hf = harmonic_frequency<MODE>( source.f, h + 1 );                 // :393
		const int hb = to_int_sat( frequency_to_bin( hf, sr, dft ) );                 // :394
		if( hb >= bins ) return false;                                                // :395
		const float mag = source.m * sp[h];                                           // :398
		if( hb >= 0 && mag > 0.0f )                                                   // :399: beats the initial 0 or a quieter occupant; false for NaN
			atomicMax( &keys[hb], ( (unsigned long long) __float_as_uint( mag ) << 32 ) | (unsigned long long) ( 0xFFFFFFFFu - unsigned( b * H + h ) ) );
		return true;
		};
	const int wide = MODE == 1 ? min( kWideBins, bins ) : 0;
	for( int b = threadIdx.x >> 6; b < wide; b += 4 )                                 // one wavefront per low bin, lanes over its overtones
		{
		const MFd source = ip[b];
		if( !( source.f <= 1.0f ) )                                                   // :389
			for( int h = threadIdx.x & 63; h < H; h += 64 ) if( !offer( b, source, h ) ) break;   // overtone bins never decrease with h
		}
	for( int b = wide + threadIdx.x; b < bins; b += 256 )
		{
		const MFd source = ip[b];
		if( source.f <= 1.0f ) continue;
		for( int h = 0; h < H; ++h ) if( !offer( b, source, h ) ) break;
		}
	__syncthreads();
	MFd * op = out + row * bins;
	for( int b = threadIdx.x; b < bins; b += 256 )
		{
		const unsigned long long key = keys[b];
		MFd r{ 0.0f, 0.0f };
		if( key )
			{
			const unsigned seq = 0xFFFFFFFFu - unsigned( key & 0xFFFFFFFFull );
			const int sb = int( seq / unsigned( H ) ), h = int( seq % unsigned( H ) );
			r = MFd{ __uint_as_float( unsigned( key >> 32 ) ), harmonic_frequency<MODE>( ip[sb].f, h + 1 ) };   // :400
			}
		op[b] = r;
		}
	}

// PV/PVModify.cpp:566-603 smear_time: one thread per output MF averages the MFs of its bin over the frames around it, weighted by
// the sampled distribution; fp32 products, fp64 sums, like the reference.  Lanes are adjacent bins, so every step of the walk over
// frames is one coalesced row segment.  grid = ( ceil(bins/256), out_frames, channels ) folded into x.
__global__ __launch_bounds__( 256 ) void k_smear_time( const MFd * in, int64_t F, int bins, float sr, float hop, const float * smear, float smear_const,
	const int * gran, int gran_const, const float * dist, int n_dist, int true_left, int64_t Fo, int blocks_per_row, MFd * out )
	{
	const int64_t row = blockIdx.x / blocks_per_row;                                  // channel * Fo + out_frame
	const int b = int( blockIdx.x % blocks_per_row ) * 256 + threadIdx.x;
	if( b >= bins ) return;
	const int64_t of = row % Fo, channel = row / Fo;
	const int64_t in_frame = min( max( of + true_left, int64_t( 0 ) ), F - 1 );       // :569
	float width_s = smear ? smear[in_frame * bins + b] : smear_const;
	width_s = width_s < 0.0f ? 0.0f : width_s;                         // :524
	const int expansion = n_dist > 0 ? max( to_int_sat( time_to_frame( width_s, sr, hop ) ), 0 ) : 0;   // :574 (a NaN size spreads over no frames)
	const int step = max( gran ? gran[in_frame * bins + b] : gran_const, 1 );   // :579, :521
	double mag_sum = 0, freq_sum = 0, weight_all = 0, weight_inside = 0;
	const MFd * col = in + channel * F * bins + b;
	for( int64_t off = -int64_t( expansion ); off < expansion; off += step )
		{
		const float rel_pos = frame_to_time( float( int( off ) ), sr, hop ) / width_s;      // :583
		int access = to_int_sat( float( n_dist ) * 0.5f * ( 1.0f + rel_pos ) );                  // :584
		access = min( max( access, 0 ), n_dist - 1 );                                                // :585
		const float w_tap = dist[access];
		weight_all += double( w_tap );
		const int64_t source = of + true_left + off;                                  // :589
		if( source < 0 || source >= F ) continue;
		const MFd src_mf = col[source * bins];
		weight_inside += double( w_tap );
		mag_sum += double( src_mf.m * w_tap );                                         // :594-595
		freq_sum += double( src_mf.f * w_tap );
		}
	if( weight_all > 0.0 ) mag_sum /= weight_all;                       // :600-601
	if( weight_inside > 0.0 ) freq_sum /= weight_inside;
	out[row * bins + b] = MFd{ float( mag_sum ), float( freq_sum ) };
	}

// ---------------------------------------------------------------------------------------------------------------------
// PV::modify (PV/PVModify.cpp:15-193): every input quad is mapped to a quad of the output and rasterised there; an output point
// keeps the loudest weighted corner offered to it.  Two passes over the output, which doubles as the key store (an MF and a key
// are both 8 bytes):
//   k_modify_offer    thread per input quad: walks its bounding box in the reference's order (x outside, y inside, `break` leaves
//                     the column) and offers ( weight bits << 32 | ~quad index ) to every point it covers, global atomic max
//   k_modify_resolve  thread per output point: decodes the winning quad and which of its corners was the loudest (two more bits of the
//                     key) and replaces the key by { weight, that corner's mapped frequency }
// The arithmetic is what g++ makes of the reference's expressions (see oracle/arrange_oracle.cpp): sqrt and the division after it in
// double, rounded to float once.  Equally loud candidates: the first quad in ( frame, bin ) order (unspecified in the reference,
// which runs frames in parallel under a mutex per output frame).
// ---------------------------------------------------------------------------------------------------------------------
struct Quad { float px[4], py[4], pm[4]; };

// 0: not inside; 1: `break`; 2: candidate
__device__ __forceinline__ int quad_point( const Quad & q, int xi, int yi, int interp_kind, float & weight, int & corner )
	{
	const float * px = q.px; const float * py = q.py;
	const float x = float( xi ), y = float( yi );
	const float e01x = px[1] - px[0], e01y = py[1] - py[0];                            // :87-90
	const float e12x = px[2] - px[1], e12y = py[2] - py[1];
	const float e23x = px[3] - px[2], e23y = py[3] - py[2];
	const float e30x = px[0] - px[3], e30y = py[0] - py[3];
	bool c = false;                                                                   // :105-109
	if( ( ( py[0] <= y && y < py[3] ) || ( py[3] <= y && y < py[0] ) ) && ( x < e30x / e30y * ( y - py[0] ) + px[0] ) ) c = !c;
	if( ( ( py[1] <= y && y < py[0] ) || ( py[0] <= y && y < py[1] ) ) && ( x < e01x / e01y * ( y - py[1] ) + px[1] ) ) c = !c;
	if( ( ( py[2] <= y && y < py[1] ) || ( py[1] <= y && y < py[2] ) ) && ( x < e12x / e12y * ( y - py[2] ) + px[2] ) ) c = !c;
	if( ( ( py[3] <= y && y < py[2] ) || ( py[2] <= y && y < py[3] ) ) && ( x < e23x / e23y * ( y - py[3] ) + px[3] ) ) c = !c;
	if( !c ) return 0;
	const float a0 = px[0], a1 = px[1] - px[0], a2 = px[3] - px[0], a3 = px[0] - px[1] + px[2] - px[3];   // :116-117
	const float b0 = py[0], b1 = py[1] - py[0], b2 = py[3] - py[0], b3 = py[0] - py[1] + py[2] - py[3];
	const float qa = a3 * b2 - a2 * b3;                                            // :119-124
	const float qb = a3 * b0 - a0 * b3 + a1 * b2 - a2 * b1 + x * b3 - a3 * y;
	const float qc = a1 * b0 - a0 * b1 + x * b1 - a1 * y;
	float m;
	if( qa == 0.0f )                                                               // :126-138
		{
		if( qb == 0.0f ) return 1;
		m = -qc / qb;
		}
	else
		{
		const float disc = qb * qb - 4.0f * qa * qc;
		if( disc < 0.0f ) return 1;
		m = float( ( double( -qb ) + sqrt( double( disc ) ) ) / double( 2.0f * qa ) );
		}
	const float den_l = a1 + a3 * m;                                           // :139-141
	if( den_l == 0.0f ) return 1;
	const float l = ( x - a0 - a2 * m ) / den_l;
	const float epsilon = 0.0001f;                                                    // :144-145
	if( fabsf( l - 0.5f ) > 0.5f + epsilon || fabsf( m - 0.5f ) > 0.5f + epsilon ) return 1;
	const float interpL = interpolate( interp_kind, l ), interpM = interpolate( interp_kind, m );   // :147-148
	const float w[4] = { ( 1.0f - interpL ) * ( 1.0f - interpM ) * q.pm[0], ( interpL ) * ( 1.0f - interpM ) * q.pm[1],   // :150-154
	                     ( interpL ) * ( interpM ) * q.pm[2], ( 1.0f - interpL ) * ( interpM ) * q.pm[3] };
	const float weight_sum = w[0] + w[1] + w[2] + w[3];                              // :155-156
	if( weight_sum <= 0.0f ) return 1;
	int largest = 0;                                                                  // std::max_element, :169-170
	float best = w[0];
	#pragma unroll
	for( int i = 1; i < 4; ++i ) if( best < w[i] ) { best = w[i]; largest = i; }
	weight = best;
	corner = largest;
	return 2;
	}

__device__ __forceinline__ Quad load_quad( const MFd * in_channel, const float2 * mod, int64_t frame, int bin, int bins, float sr, float hop, float dft )
	{
	const int64_t idx[4] = { ( frame - 1 ) * bins + bin - 1, frame * bins + bin - 1, frame * bins + bin, ( frame - 1 ) * bins + bin };   // :76-86
	Quad q;
	#pragma unroll
	for( int k = 0; k < 4; ++k )
		{
		const float2 v = mod[idx[k]];
		q.px[k] = time_to_frame( v.x, sr, hop );                                      // :23-26
		q.py[k] = frequency_to_bin( v.y, sr, dft );
		q.pm[k] = in_channel[idx[k]].m;
		}
	return q;
	}

__global__ __launch_bounds__( 256 ) void k_modify_offer( const MFd * in, int64_t F, int bins, float sr, float hop, float dft, const float2 * mod, int interp_kind,
	int64_t Fo, int blocks_per_row, unsigned long long * out_keys )
	{
	const int64_t row = blockIdx.x / blocks_per_row;                                  // channel * ( F - 1 ) + ( frame - 1 )
	const int bin = 1 + int( blockIdx.x % blocks_per_row ) * 256 + threadIdx.x;
	if( bin >= bins ) return;
	const int64_t frame = 1 + row % ( F - 1 ), channel = row / ( F - 1 );
	const Quad q = load_quad( in + channel * F * bins, mod, frame, bin, bins, sr, hop, dft );
	// A quad with a NaN or infinite corner is skipped: NaNs then run through every expression of :116-156 and no candidate survives
	// them (except through the interpolators that ignore their argument -- an accident not reproduced), while the bounding box of
	// such a quad is the whole output.
	bool finite = true;
	#pragma unroll
	for( int k = 0; k < 4; ++k ) finite = finite && fabsf( q.px[k] ) < INFINITY && fabsf( q.py[k] ) < INFINITY;
	if( !finite ) return;
	// :93-96: the bounding box, clipped to the output
	const int minx = to_int_sat( fmaxf( floorf( fminf( fminf( q.px[0], q.px[1] ), fminf( q.px[2], q.px[3] ) ) ), 0.0f ) );
	const int miny = to_int_sat( fmaxf( floorf( fminf( fminf( q.py[0], q.py[1] ), fminf( q.py[2], q.py[3] ) ) ), 0.0f ) );
	// ( the min in integers as well: float( Fo - 1 ) may round up once Fo passes 2^24 )
	const int maxx = min( to_int_sat( fminf( ceilf( fmaxf( fmaxf( q.px[0], q.px[1] ), fmaxf( q.px[2], q.px[3] ) ) ), float( Fo - 1 ) ) ), int( Fo - 1 ) );
	const int maxy = min( to_int_sat( fminf( ceilf( fmaxf( fmaxf( q.py[0], q.py[1] ), fmaxf( q.py[2], q.py[3] ) ) ), float( bins - 1 ) ) ), bins - 1 );
	const unsigned seq4 = unsigned( frame * bins + bin ) << 2;                        // the loudest corner rides in the two low bits
	unsigned long long * keys = out_keys + channel * Fo * bins;
	for( int x = minx; x <= maxx; ++x )                                               // :99-101
		for( int y = miny; y <= maxy; ++y )
			{
			float weight; int corner;
			const int r = quad_point( q, x, y, interp_kind, weight, corner );
			if( r == 0 ) continue;
			if( r == 1 ) break;
			if( weight > 0.0f )                                                       // :175: beats the cleared output or a quieter occupant; false for NaN
				atomicMax( &keys[int64_t( x ) * bins + y], ( (unsigned long long) __float_as_uint( weight ) << 32 ) | (unsigned long long) ( 0xFFFFFFFFu - ( seq4 | unsigned( corner ) ) ) );
			}
	}

__global__ __launch_bounds__( 256 ) void k_modify_resolve( int64_t F, int bins, const float * in_f, int64_t Fo, int blocks_per_row, MFd * out )
	{
	const int64_t row = blockIdx.x / blocks_per_row;                                  // channel * Fo + x
	const int y = int( blockIdx.x % blocks_per_row ) * 256 + threadIdx.x;
	if( y >= bins ) return;
	const int64_t channel = row / Fo;
	const unsigned long long key = reinterpret_cast<const unsigned long long *>( out )[row * bins + y];
	if( key == 0ull ) return;                                                         // nothing offered: stays { 0, 0 }
	const unsigned packed = 0xFFFFFFFFu - unsigned( key & 0xFFFFFFFFull );
	const unsigned seq = packed >> 2;
	const int corner = int( packed & 3u );
	const int64_t frame = seq / unsigned( bins );
	const int bin = int( seq % unsigned( bins ) );
	const int64_t idx = ( frame - ( corner == 0 || corner == 3 ? 1 : 0 ) ) * bins + bin - ( corner < 2 ? 1 : 0 );   // :76-86: corners 0..3 of the quad
	out[row * bins + y] = MFd{ __uint_as_float( unsigned( key >> 32 ) ), in_f[channel * F * bins + idx] };   // :176
	}

// ---------------------------------------------------------------------------------------------------------------------
// PV::stretch_spline (PV/PVModify.cpp:387-443): a natural cubic spline through every (channel, bin) column's magnitudes and one
// through its frequencies, knots where the caller's per-frame steps put the input frames, evaluated on every output frame; fp64,
// in the operation order of the spline the reference vendors (spline/spline.h:284-401 with band_matrix's LU, :187-261).
// The tridiagonal matrix depends on the knots only: its LU factors are computed once on the host (flanhip_stretch_spline_dev) and
// shared by all columns.  What remains per column is two first-order recurrences over the frames (L y = rhs forward, R b = y
// backward) and the evaluation: one thread per ( channel, bin, m | f ), adjacent threads adjacent floats of a frame, so every step of
// the walk moves whole coalesced rows; the intermediate vector lives in a workspace [frame][column].
// ---------------------------------------------------------------------------------------------------------------------
struct SplineFactors { const double * x, * sd, * lo, * up, * di; };       // knots; 1 / a_ii; L below the diagonal; R above it; R's diagonal

// Only the two substitutions are sequential in the frame; the right-hand side before them and the coefficients and the evaluation
// after them are independent per ( frame, column ) and run as ordinary wide kernels.

// rhs( i ) * saved_diag( i ), spline.h:306 and the first factor of :233.  grid = ( ceil(row_floats/256) * F, channels )
__global__ __launch_bounds__( 256 ) void k_spline_rhs( const float * in, int64_t F, int row_floats, SplineFactors fac, double * ws, int64_t ncols, int col_blocks )
	{
	const int col_in_channel = int( blockIdx.x % col_blocks ) * 256 + threadIdx.x;
	if( col_in_channel >= row_floats ) return;
	const int64_t i = blockIdx.x / col_blocks, channel = blockIdx.y;
	double rhs = 0.0;
	if( i > 0 && i + 1 < F )                                                          // 0 in the first and last row, :313,:327
		{
		const float * yp = in + ( channel * F + i ) * row_floats + col_in_channel;
		const double y_prev = double( yp[-row_floats] ), y_cur = double( yp[0] ), y_next = double( yp[row_floats] );
		const double x_prev = fac.x[i - 1], x_cur = fac.x[i], x_next = fac.x[i + 1];
		rhs = ( y_next - y_cur ) / ( x_next - x_cur ) - ( y_cur - y_prev ) / ( x_cur - x_prev );
		}
	ws[i * ncols + channel * row_floats + col_in_channel] = rhs * fac.sd[i];
	}

// L y = rhs forward (spline.h:222-235), R b = y backward (:237-250): one thread per column, kSolveLanes columns per wavefront (the
// walk is bound by the latency of its dependent fp64 steps, not by lanes: narrow wavefronts put a walk on every SIMD of the chip).
// Batches of B frames: the loads of a batch are issued first (left in the loop they queue up behind the previous step's store).
constexpr int kSolveLanes = 16;
__global__ __launch_bounds__( 64 ) void k_spline_solve( int64_t F, SplineFactors fac, double * ws, int64_t ncols )
	{
	if( threadIdx.x >= kSolveLanes ) return;
	const int64_t col = int64_t( blockIdx.x ) * kSolveLanes + threadIdx.x;
	if( col >= ncols ) return;
	double * wp = ws + col;                                                           // w( i ) = wp[i * ncols]
	constexpr int B = 16;
		{
		double yt_prev = 0.0;
		for( int64_t base = 0; base < F; base += B )
			{
			double w[B], los[B];
			#pragma unroll
			for( int k = 0; k < B; ++k ) { const int64_t i = base + k; w[k] = i < F ? wp[i * ncols] : 0.0; los[k] = i < F ? fac.lo[i] : 0.0; }
			#pragma unroll
			for( int k = 0; k < B; ++k )
				{
				double sum = 0;
				if( base + k > 0 ) sum += los[k] * yt_prev;
				w[k] = w[k] - sum;
				yt_prev = w[k];
				}
			#pragma unroll
			for( int k = 0; k < B; ++k ) if( base + k < F ) wp[( base + k ) * ncols] = w[k];
			}
		}
		{
		double b_next = 0.0;
		for( int64_t top = F - 1; top >= 0; top -= B )
			{
			double w[B], ups[B], dis[B];
			#pragma unroll
			for( int k = 0; k < B; ++k ) { const int64_t i = top - k; w[k] = i >= 0 ? wp[i * ncols] : 0.0; ups[k] = i >= 0 ? fac.up[i] : 0.0; dis[k] = i >= 0 ? fac.di[i] : 1.0; }
			#pragma unroll
			for( int k = 0; k < B; ++k )
				{
				double sum = 0;
				if( top - k < F - 1 ) sum += ups[k] * b_next;
				w[k] = ( w[k] - sum ) / dis[k];
				b_next = w[k];
				}
			#pragma unroll
			for( int k = 0; k < B; ++k ) if( top - k >= 0 ) wp[( top - k ) * ncols] = w[k];
			}
		}
	}

// coefficients of a segment (:343-349) and the output frames it holds: operator() (:375-397) puts frame t into the segment that
// starts at the last knot strictly below t (lower_bound), frame 0 into the first.  grid = ( ceil(row_floats/256) * (F-1), channels )
__global__ __launch_bounds__( 256 ) void k_spline_eval( const float * in, int64_t F, int row_floats, SplineFactors fac, const double * ws, int64_t ncols, int64_t Fo, int col_blocks, float * out )
	{
	const int col_in_channel = int( blockIdx.x % col_blocks ) * 256 + threadIdx.x;
	if( col_in_channel >= row_floats ) return;
	const int64_t i = blockIdx.x / col_blocks, channel = blockIdx.y;
	constexpr double third = 1.0 / 3.0;
	const float * yp = in + ( channel * F + i ) * row_floats + col_in_channel;
	const double * wp = ws + i * ncols + channel * row_floats + col_in_channel;
	const double y0 = double( yp[0] ), y1 = double( yp[row_floats] ), b0 = wp[0], b1 = wp[ncols];
	const double x0 = fac.x[i], x1 = fac.x[i + 1], dx = x1 - x0;
	const double a = third * ( b1 - b0 ) / dx;
	const double c = ( y1 - y0 ) / dx - third * ( 2.0 * b0 + b1 ) * dx;
	float * op = out + channel * Fo * row_floats + col_in_channel;
	const int64_t t_first = i == 0 ? 0 : int64_t( x0 ) + 1, t_last = min( int64_t( x1 ), Fo - 1 );
	for( int64_t t = t_first; t <= t_last; ++t )
		{
		const double h = double( t ) - x0;
		op[t * row_floats] = float( ( ( a * h + b0 ) * h + c ) * h + y0 );
		}
	}

} // namespace flanhip

namespace flanhip { int processors_arrange_set_interp_lut( int slot, const float * d_table ) { return set_interp_lut_here( slot, d_table ); } }

using namespace flanhip;

extern "C" {

int flanhip_get_frame_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float frame_pos, int interp_kind, flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	FLANHIP_REQUIRE( frame_pos >= 0.0f && frame_pos <= float( F - 1 ), FLANHIP_ERR_INVALID_ARG, "frame position outside [0, F-1] (PV.cpp:28 clamps it)" );
	FLANHIP_REQUIRE( valid_interp( interp_kind ), FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( ch <= 65535, FLANHIP_ERR_INVALID_ARG, "too many channels" );
	const float lo = std::floor( frame_pos ), hi = std::ceil( frame_pos );
	const float mix = interp_eval_host( interp_kind, frame_pos - lo );                // PV.cpp:67: the interpolator of [0,1) -> [0,1], one value: on the host
	hipLaunchKernelGGL( k_get_frame, dim3( ( bins + 255 ) / 256, unsigned( ch ) ), dim3( 256 ), 0, (hipStream_t) stream, (const MFd*) d_pv, F, bins,
		int64_t( lo ), int64_t( hi ), mix, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

static int select_frames( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const int32_t * d_src, int64_t start, int64_t Fo, flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	FLANHIP_REQUIRE( Fo > 0 && ch * Fo < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_INVALID_ARG, "bad output frame count" );
	hipLaunchKernelGGL( k_select_frames, dim3( unsigned( ch * Fo ) ), dim3( 256 ), 0, (hipStream_t) stream, (const MFd*) d_pv, F, bins, d_src, start, Fo, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_select_frames_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const int32_t * d_src_frames, int64_t out_frames, flanhip_MF * d_out, void * stream )
	{
	FLANHIP_REQUIRE( d_src_frames, FLANHIP_ERR_INVALID_ARG, "null frame map" );
	return select_frames( d_pv, ch, F, bins, d_src_frames, 0, out_frames, d_out, stream );
	}

int flanhip_cut_frames_range( int64_t F, int32_t start, int32_t end, int32_t * start_out, int32_t * count_out )
	{
	FLANHIP_REQUIRE( start_out && count_out, FLANHIP_ERR_INVALID_ARG, "null out pointer" );
	*start_out = 0; *count_out = 0;
	if( end <= start || F <= 0 ) return FLANHIP_OK;                                   // PV.cpp:651: a null PV
	start = std::clamp( start, int32_t( 0 ), int32_t( F - 1 ) );                      // :652-653
	end = std::clamp( end, int32_t( 0 ), int32_t( F - 1 ) );
	*start_out = start; *count_out = std::max( end - start, 0 );
	return FLANHIP_OK;
	}

int flanhip_cut_frames_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, int64_t start, int64_t count, flanhip_MF * d_out, void * stream )
	{
	FLANHIP_REQUIRE( start >= 0 && count > 0 && start + count <= F, FLANHIP_ERR_INVALID_ARG, "frame range outside the PV" );
	return select_frames( d_pv, ch, F, bins, nullptr, start, count, d_out, stream );
	}

int64_t flanhip_freeze_plan( int64_t F, float sr, int hop, const float * times, const float * lengths, int n, int32_t * src_frames )
	{
	if( F <= 0 || F > INT32_MAX || hop < 1 || !( sr > 0.0f ) || n < 0 || ( n > 0 && ( !times || !lengths ) ) ) return -1;
	auto to_frame = [&]( float seconds )                                              // Frame( time_to_frame( t ) ), PVBuffer.cpp:428-431, saturating
		{
		const float v = seconds * sr / float( hop );
		if( !( v == v ) ) return INT32_MIN;
		if( v >= 2147483648.0f ) return INT32_MAX;
		if( v <= -2147483648.0f ) return INT32_MIN;
		return int32_t( v );
		};
	struct Event { int32_t frame, length; };
	std::vector<Event> ev( static_cast<size_t>( n ) );
	for( int i = 0; i < n; ++i )                                                      // PV.cpp:150-156
		ev[size_t( i )] = Event{ std::clamp( to_frame( times[i] ), int32_t( 0 ), int32_t( F - 1 ) ), std::max( to_frame( lengths[i] ), int32_t( 0 ) ) };
	// :159-165: sorted by frame, one event per frame.  The reference's sort is not stable, so WHICH of several events on one frame
	// survives is unspecified there; here it is the first one given.
	std::stable_sort( ev.begin(), ev.end(), []( const Event & a, const Event & b ){ return a.frame < b.frame; } );
	ev.erase( std::unique( ev.begin(), ev.end(), []( const Event & a, const Event & b ){ return a.frame == b.frame; } ), ev.end() );
	float total = 0;                                                                  // :167-168: a float
	for( const Event & e : ev ) total += float( e.length );
	const float frames_f = float( int32_t( F ) ) + total;                             // :170-171
	if( !( frames_f < 2147483648.0f ) ) return -1;
	const int64_t Fo = int64_t( frames_f );
	if( !src_frames ) return Fo;
	std::fill( src_frames, src_frames + Fo, int32_t( -1 ) );                          // frames the loops never reach stay zero
	size_t k = 0;
	int64_t o = 0;
	for( int64_t in = 0; in < F; ++in )                                               // :176-195
		{
		if( k < ev.size() && in == ev[k].frame )
			{
			for( int32_t r = 0; r < ev[k].length; ++r, ++o ) if( o < Fo ) src_frames[o] = int32_t( in );
			++k;
			}
		else { if( o < Fo ) src_frames[o] = int32_t( in ); ++o; }
		}
	return Fo;
	}

int flanhip_place_frames_dev( const flanhip_MF * d_in, int64_t in_ch, int64_t in_F, int in_bins, flanhip_MF * d_out, int64_t out_ch, int64_t out_F, int out_bins,
	int64_t out_start, void * stream )
	{
	FLANHIP_REQUIRE( d_in && d_out, FLANHIP_ERR_INVALID_ARG, "null buffer" );
	FLANHIP_REQUIRE( in_ch > 0 && in_F > 0 && in_bins > 0 && out_ch > 0 && out_F > 0 && out_bins > 0, FLANHIP_ERR_INVALID_ARG, "bad sizes" );
	FLANHIP_REQUIRE( out_start >= 0 && out_start + in_F <= out_F, FLANHIP_ERR_INVALID_ARG, "frames do not fit the output" );
	if( int rc = require_device() ) return rc;
	const int64_t rows = std::min( in_ch, out_ch ) * in_F;
	FLANHIP_REQUIRE( rows < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_INVALID_ARG, "too many rows" );
	// rows of channels the output lacks are not visited: the grid covers min( in_ch, out_ch ) channels, which are the first ones
	hipLaunchKernelGGL( k_place_frames, dim3( unsigned( rows ) ), dim3( 256 ), 0, (hipStream_t) stream, (const MFd*) d_in, in_F, in_bins, (MFd*) d_out, out_F, out_bins,
		out_start, std::min( in_bins, out_bins ) );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_select_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int hop, const float * d_selector_tf, int64_t out_frames,
	flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( d_selector_tf && hop >= 1, FLANHIP_ERR_INVALID_ARG, "null selector grid or bad hop" );
	FLANHIP_REQUIRE( out_frames > 0 && ch * out_frames < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_INVALID_ARG, "bad output frame count" );
	hipLaunchKernelGGL( k_select, dim3( unsigned( ch * out_frames ) ), dim3( 256 ), 0, (hipStream_t) stream, (const MFd*) d_pv, F, bins, sr, float( hop ),
		float( ( bins - 1 ) * 2 ), reinterpret_cast<const float2*>( d_selector_tf ), out_frames, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_harmonic_scale_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, const float * d_series, int num_harmonics, int mode,
	flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( mode == 0 || mode == 1, FLANHIP_ERR_INVALID_ARG, "mode: 0 octaves, 1 harmonics" );
	FLANHIP_REQUIRE( ch * F < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_INVALID_ARG, "too many rows" );
	if( num_harmonics <= 0 )                                                          // no overtone is ever placed: the cleared output (PV.cpp:369-370)
		{
		FLANHIP_CHECK( hipMemsetAsync( d_out, 0, sizeof( MFd ) * size_t( ch ) * F * bins, (hipStream_t) stream ) );
		return FLANHIP_OK;
		}
	FLANHIP_REQUIRE( d_series, FLANHIP_ERR_INVALID_ARG, "null series grid" );
	FLANHIP_REQUIRE( int64_t( bins ) * num_harmonics < ( int64_t( 1 ) << 32 ), FLANHIP_ERR_UNSUPPORTED, "bins x harmonics does not fit the placement key" );
	const size_t key_bytes = sizeof( unsigned long long ) * size_t( bins );
	FLANHIP_REQUIRE( key_bytes <= 65536, FLANHIP_ERR_UNSUPPORTED, "dft sizes above 8192 are not supported by add_octaves / add_harmonics" );
	const size_t staged_bytes = key_bytes + sizeof( MFd ) * size_t( bins ) + sizeof( float ) * size_t( num_harmonics );
	const int staged = staged_bytes <= 65536 ? 1 : 0;                                 // input row and series row next to the keys when they fit
	const size_t lds = staged ? staged_bytes : key_bytes;
	if( mode == 0 )
		hipLaunchKernelGGL( k_harmonic_scale<0>, dim3( unsigned( ch * F ) ), dim3( 256 ), lds, (hipStream_t) stream, (const MFd*) d_pv, F, bins, sr, float( ( bins - 1 ) * 2 ),
			d_series, num_harmonics, staged, (MFd*) d_out );
	else
		hipLaunchKernelGGL( k_harmonic_scale<1>, dim3( unsigned( ch * F ) ), dim3( 256 ), lds, (hipStream_t) stream, (const MFd*) d_pv, F, bins, sr, float( ( bins - 1 ) * 2 ),
			d_series, num_harmonics, staged, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int64_t flanhip_modify_out_frames( const float * mod_tf, int64_t F, int bins, float sr, int hop )
	{
	if( !mod_tf || F <= 0 || bins <= 0 || hop < 1 || !( sr > 0.0f ) ) return -1;
	auto to_frame = [&]( float t ){ return t * sr / float( hop ); };                  // time_to_frame, PVBuffer.cpp:428-431
	float mx = to_frame( mod_tf[0] );
	for( int64_t i = 1; i < F * bins; ++i ) { const float v = to_frame( mod_tf[2 * i] ); if( mx < v ) mx = v; }   // PVModify.cpp:29: ranges::max_element, projected
	const float last = std::ceil( mx );
	if( last / ( sr / float( hop ) ) > 60.0f * 10.0f ) return -2;                     // :31: "longer than 10 minutes, which is currently disabled"
	if( !( last == last ) || last < 1.0f ) return 0;
	if( last >= 2147483648.0f ) return -1;
	return int64_t( std::ceil( last ) );                                              // :38
	}

int flanhip_modify_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int hop, const float * d_mod_tf, const float * d_in_f, int interp_kind,
	int64_t out_frames, flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( d_mod_tf && d_in_f && hop >= 1 && out_frames > 0, FLANHIP_ERR_INVALID_ARG, "null grid, bad hop or bad output frame count" );
	FLANHIP_REQUIRE( valid_interp( interp_kind ), FLANHIP_ERR_INVALID_ARG, "unknown interpolator" );
	FLANHIP_REQUIRE( F * bins < ( int64_t( 1 ) << 30 ), FLANHIP_ERR_UNSUPPORTED, "frames x bins does not fit the quad index" );
	hipStream_t s = (hipStream_t) stream;
	FLANHIP_CHECK( hipMemsetAsync( d_out, 0, sizeof( MFd ) * size_t( ch ) * out_frames * bins, s ) );   // PVModify.cpp:40
	if( F < 2 ) return FLANHIP_OK;                                                    // no quads
	const float dft = float( ( bins - 1 ) * 2 );
	const int quad_blocks = ( bins - 1 + 255 ) / 256, point_blocks = ( bins + 255 ) / 256;
	FLANHIP_REQUIRE( ch * ( F - 1 ) * quad_blocks < ( int64_t( 1 ) << 31 ) && ch * out_frames * point_blocks < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_INVALID_ARG, "too many rows" );
	hipLaunchKernelGGL( k_modify_offer, dim3( unsigned( ch * ( F - 1 ) * quad_blocks ) ), dim3( 256 ), 0, s, (const MFd*) d_pv, F, bins, sr, float( hop ), dft,
		reinterpret_cast<const float2*>( d_mod_tf ), interp_kind, out_frames, quad_blocks, reinterpret_cast<unsigned long long*>( d_out ) );
	FLANHIP_CHECK( hipGetLastError() );
	hipLaunchKernelGGL( k_modify_resolve, dim3( unsigned( ch * out_frames * point_blocks ) ), dim3( 256 ), 0, s, F, bins, d_in_f, out_frames, point_blocks, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int64_t flanhip_stretch_spline_out_frames( const uint32_t * steps, int64_t F )
	{
	if( !steps || F < 3 ) return -1;                                                  // spline.h:288: more than two knots
	int64_t total = 0;
	for( int64_t i = 0; i + 1 < F; ++i ) { if( steps[i] < 1 ) return -1; total += steps[i]; }   // PVModify.cpp:391-394: every step is at least 1
	return total <= INT32_MAX ? total : -1;                                           // :399-405
	}

int flanhip_stretch_spline_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, const uint32_t * steps, int64_t out_frames, flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, 1.0f ) ) return rc;
	FLANHIP_REQUIRE( steps && F >= 3, FLANHIP_ERR_INVALID_ARG, "stretch_spline needs at least three frames (spline.h:288) and the per-frame steps" );
	FLANHIP_REQUIRE( out_frames == flanhip_stretch_spline_out_frames( steps, F ), FLANHIP_ERR_INVALID_ARG, "out_frames is not the sum of the steps (each >= 1)" );
	FLANHIP_REQUIRE( ch <= 65535 && int64_t( ( bins * 2 + 255 ) / 256 ) * F < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_UNSUPPORTED, "stretch_spline: too many channels or frames" );
	// the knots and the LU factors of the spline's tridiagonal system (spline.h:298-327, :187-219): the same for every column
	const size_t n = size_t( F );
	std::vector<double> host( 5 * n );
	double * x = host.data(), * sd = x + n, * lo = sd + n, * up = lo + n, * di = up + n;
	x[0] = 0.0;
	for( size_t i = 0; i + 1 < n; ++i ) x[i + 1] = x[i] + double( steps[i] );           // PVModify.cpp:400-405 (integers: exact)
	for( size_t i = 1; i + 1 < n; ++i )                                               // spline.h:302-306
		{
		lo[i] = 1.0 / 3.0 * ( x[i] - x[i - 1] );
		di[i] = 2.0 / 3.0 * ( x[i + 1] - x[i - 1] );
		up[i] = 1.0 / 3.0 * ( x[i + 1] - x[i] );
		}
	lo[0] = 0.0; di[0] = 2.0; up[0] = 0.0;                                            // :311-312
	di[n - 1] = 2.0; lo[n - 1] = 0.0; up[n - 1] = 0.0;                                // :325-326
	for( size_t i = 0; i < n; ++i )                                                   // :194-204
		{
		sd[i] = 1.0 / di[i];
		if( i > 0 ) lo[i] *= sd[i];
		if( i + 1 < n ) up[i] *= sd[i];
		di[i] = 1.0;
		}
	for( size_t k = 0; k + 1 < n; ++k )                                               // :207-219
		{
		const double xk = -lo[k + 1] / di[k];
		lo[k + 1] = -xk;
		di[k + 1] = di[k + 1] + xk * up[k];
		}
	hipStream_t s = (hipStream_t) stream;
	const int row_floats = bins * 2;
	const int64_t ncols = ch * row_floats;
	double * d_fac = nullptr, * d_ws = nullptr;
	retain_pool_memory();
	FLANHIP_CHECK( hipMallocAsync( reinterpret_cast<void**>( &d_fac ), sizeof( double ) * host.size(), s ) );
	FLANHIP_CHECK( hipMallocAsync( reinterpret_cast<void**>( &d_ws ), sizeof( double ) * n * size_t( ncols ), s ) );
	FLANHIP_CHECK( hipMemcpyAsync( d_fac, host.data(), sizeof( double ) * host.size(), hipMemcpyHostToDevice, s ) );
	FLANHIP_CHECK( hipStreamSynchronize( s ) );                                       // `host` dies with this call
	const SplineFactors fac{ d_fac, d_fac + n, d_fac + 2 * n, d_fac + 3 * n, d_fac + 4 * n };
	const int64_t col_blocks = ( row_floats + 255 ) / 256;
	hipLaunchKernelGGL( k_spline_rhs, dim3( unsigned( col_blocks * F ), unsigned( ch ) ), dim3( 256 ), 0, s, reinterpret_cast<const float*>( d_pv ), F, row_floats, fac, d_ws, ncols,
		int( col_blocks ) );
	FLANHIP_CHECK( hipGetLastError() );
	hipLaunchKernelGGL( k_spline_solve, dim3( unsigned( ( ncols + kSolveLanes - 1 ) / kSolveLanes ) ), dim3( 64 ), 0, s, F, fac, d_ws, ncols );
	FLANHIP_CHECK( hipGetLastError() );
	hipLaunchKernelGGL( k_spline_eval, dim3( unsigned( col_blocks * ( F - 1 ) ), unsigned( ch ) ), dim3( 256 ), 0, s, reinterpret_cast<const float*>( d_pv ), F, row_floats, fac, d_ws, ncols,
		out_frames, int( col_blocks ), reinterpret_cast<float*>( d_out ) );
	FLANHIP_CHECK( hipGetLastError() );
	FLANHIP_CHECK( hipFreeAsync( d_ws, s ) );
	FLANHIP_CHECK( hipFreeAsync( d_fac, s ) );
	return FLANHIP_OK;
	}

int flanhip_smear_time_plan( int64_t F, int bins, float sr, int hop, const float * smear, float smear_const, int32_t * true_left, int64_t * out_frames, int32_t * dist_samples_2 )
	{
	FLANHIP_REQUIRE( true_left && out_frames && dist_samples_2, FLANHIP_ERR_INVALID_ARG, "null out pointer" );
	FLANHIP_REQUIRE( F > 0 && F <= INT32_MAX && bins > 0 && hop >= 1 && sr > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad sizes" );
	auto expansion_of = [&]( float v )                                                // Frame( time_to_frame( max( s, 0 ) ) ), PVModify.cpp:524,:544
		{
		v = v < 0.0f ? 0.0f : v;
		const float fr = v * sr / float( hop );
		if( !( fr == fr ) ) return int64_t( 0 );
		return int64_t( fr >= 2147483648.0f ? INT32_MAX : int32_t( fr ) );
		};
	int64_t left = 0, right = F - 1;                                                  // :536-537
	float mx = smear ? smear[0] : smear_const;
	mx = mx < 0.0f ? 0.0f : mx;
	if( smear )
		for( int64_t fr = 0; fr < F; ++fr )                                           // :538-550 (min / max over everything: the loop order is free)
			for( int b = 0; b < bins; ++b )
				{
				float v = smear[fr * bins + b];
				v = v < 0.0f ? 0.0f : v;
				const int64_t e = expansion_of( v );
				left = std::min( left, fr - e );
				right = std::max( right, fr + e );
				}
	else { const int64_t e = expansion_of( smear_const ); left = -e; right = F - 1 + e; }
	if( smear )                                                                       // FunctionSample::maximum = std::max_element, in buffer order
		for( int64_t i = 1; i < F * bins; ++i ) { float v = smear[i]; v = v < 0.0f ? 0.0f : v; if( mx < v ) mx = v; }
	FLANHIP_REQUIRE( left >= INT32_MIN && right - left <= INT32_MAX, FLANHIP_ERR_INVALID_ARG, "smear sizes beyond the frame range" );
	*true_left = int32_t( left );
	*out_frames = right - left;                                                       // :563
	const int64_t half = expansion_of( mx ) * 2;                                      // :555-556
	FLANHIP_REQUIRE( half <= INT32_MAX / 2, FLANHIP_ERR_INVALID_ARG, "smear sizes beyond the frame range" );
	*dist_samples_2 = int32_t( half );
	return FLANHIP_OK;
	}

int flanhip_smear_time_dev( const flanhip_MF * d_pv, int64_t ch, int64_t F, int bins, float sr, int hop, const float * d_smear, float smear_const,
	const int32_t * d_granularity, int32_t granularity_const, const float * d_distribution, int64_t n_distribution, int32_t true_left, int64_t out_frames,
	flanhip_MF * d_out, void * stream )
	{
	if( int rc = check_pv_args( d_pv, d_out, ch, F, bins, sr ) ) return rc;
	FLANHIP_REQUIRE( hop >= 1 && out_frames > 0, FLANHIP_ERR_INVALID_ARG, "bad hop or output frame count" );
	FLANHIP_REQUIRE( n_distribution >= 0 && n_distribution <= INT32_MAX && ( n_distribution == 0 || d_distribution ), FLANHIP_ERR_INVALID_ARG, "bad distribution table" );
	const int blocks_per_row = ( bins + 255 ) / 256;
	FLANHIP_REQUIRE( ch * out_frames * blocks_per_row < ( int64_t( 1 ) << 31 ), FLANHIP_ERR_INVALID_ARG, "too many rows" );
	// an empty table is legal only when no point spreads at all (dist_samples_2 = 0): the walk then never reads it
	hipLaunchKernelGGL( k_smear_time, dim3( unsigned( ch * out_frames * blocks_per_row ) ), dim3( 256 ), 0, (hipStream_t) stream, (const MFd*) d_pv, F, bins, sr, float( hop ),
		d_smear, smear_const, d_granularity, granularity_const, d_distribution, int( n_distribution ), true_left, out_frames, blocks_per_row, (MFd*) d_out );
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

} // extern "C"
