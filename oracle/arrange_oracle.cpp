// arrange_oracle.cpp -- CPU restatement of the PV methods that select, rearrange and re-place frames and bins:
// get_frame, select, freeze, cut_frames, join, add_octaves / add_harmonics (PV/PV.cpp:24-39, :92-127, :129-198, :362-419,
// :643-668, :698-720 of the reference; paths relative to /root/reference/src/flan).
//
// TEST INFRASTRUCTURE ONLY (same rules as flan_oracle.cpp: only tests/, smoke() and bench.py's cpu_baseline leg may use it).
//
// Pinning status: PV/PV.cpp is UNBUILDABLE here (FFTW3f, libsndfile, MSVC-only std::_Pi) and the reference ships no tests or
// vectors for these methods: PARITY UNPINNED by reference fixtures.  The restatement is literal (same loops, same fp32
// expressions, same order); where the reference is undefined the choice made is stated at the function.  User callables are
// sampled by the CALLER the way the reference samples them on the host before its loops; this file takes the grids.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct MF { float m, f; };
struct TF { float t, f; };

// PVBuffer.cpp:428-446
float bin_to_frequency( float b, float sr, int dft ) { return b * float( sr ) / float( dft ); }
float frequency_to_bin( float f, float sr, int dft ) { return f / ( float( sr ) / float( dft ) ); }
float time_to_frame( float t, float sr, int hop ) { return t * float( sr ) / float( hop ); }

// float -> Frame / Bin: undefined in C++ outside the int range; checker and device saturate, NaN -> INT_MIN
int32_t to_int( float v )
	{
	if( !( v == v ) ) return INT32_MIN;
	if( v >= 2147483648.0f ) return INT32_MAX;
	if( v <= -2147483648.0f ) return INT32_MIN;
	return int32_t( v );
	}

float interpolate( int kind, float x )                                   // Utility/Interpolator.cpp:14-101, as in processors_oracle.cpp
	{
	static const float pi = std::acos( -1.0f );
	switch( kind )
		{
		case 1: return 0.5f;
		case 2: return std::round( x );
		case 3: return 0.0f;
		case 4: return 1.0f;
		case 5: return x * x * ( 3.0f - 2.0f * x );
		case 6: return x * x * x * ( x * ( x * 6.0f - 15.0f ) + 10.0f );
		case 7: return std::sqrt( x );
		case 8: return ( 1.0f - std::cos( pi * x ) ) / 2.0f;
		case 100: return x * x;                                                      // stand-ins for a user's callable, as in processors_oracle.cpp
		case 101: return x < 0.5f ? 2.0f * x * x : 1.0f - 2.0f * ( 1.0f - x ) * ( 1.0f - x );
		case 102: return x < 0.3f ? 0.0f : 1.0f;
		}
	return x;
	}

inline size_t pos( int64_t F, int bins, int c, int64_t fr, int b ) { return ( size_t( c ) * size_t( F ) + size_t( fr ) ) * size_t( bins ) + size_t( b ); }

}

extern "C" {

// PV.cpp:24-39 with getBinInterpolated( Channel, float frame, Bin ), :62-73.  frame_pos = clamp( time_to_frame( time ), 0, F-1 )
// is the caller's (:28).  out: MF[ch][1][bins]
int oracle_get_frame( const float * pv_mf, int ch, int64_t F, int bins, float frame_pos, int interp_kind, float * out_mf )
	{
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	const int64_t lo = int64_t( std::floor( frame_pos ) ), hi = int64_t( std::ceil( frame_pos ) );
	if( lo < 0 || hi >= F ) return -1;
	const float mix = interpolate( interp_kind, frame_pos - std::floor( frame_pos ) );   // :67
	for( int c = 0; c < ch; ++c )
		for( int b = 0; b < bins; ++b )
			{
			const MF l = pv[pos( F, bins, c, lo, b )], h = pv[pos( F, bins, c, hi, b )];
			out[size_t( c ) * bins + b] = MF{ ( 1.0f - mix ) * l.m + mix * h.m, ( 1.0f - mix ) * l.f + mix * h.f };   // :69-72
			}
	return 0;
	}

// PV.cpp:92-127.  selector: TF[Fo][bins], the selector sampled over the OUTPUT's domain (:103); out: MF[ch][Fo][bins], zero where
// nothing is selected (the output is constructed zeroed).
int oracle_select( const float * pv_mf, int ch, int64_t F, int bins, float sr, int hop, const float * selector_tf, int64_t Fo, float * out_mf )
	{
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	const TF * sel = reinterpret_cast<const TF*>( selector_tf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	const int dft = ( bins - 1 ) * 2;
	std::memset( out, 0, sizeof( MF ) * size_t( ch ) * Fo * bins );
	for( int c = 0; c < ch; ++c )
		for( int64_t fr = 0; fr < Fo; ++fr )
			for( int b = 0; b < bins; ++b )
				{
				const TF s = sel[size_t( fr ) * bins + b];                                 // :110
				const int32_t sf = to_int( time_to_frame( s.t, sr, hop ) );                // :111
				const int32_t sb = to_int( frequency_to_bin( s.f, sr, dft ) );             // :112
				if( sf < 0 || F - 1 <= sf || sb < 0 || bins - 1 <= sb ) continue;          // :114-116
				MF m = pv[pos( F, bins, c, sf, sb )];
				if( s.f > 1 ) m.f *= bin_to_frequency( float( b ), sr, dft ) / s.f;        // :119-120
				out[pos( Fo, bins, c, fr, b )] = m;
				}
	return 0;
	}

// PV.cpp:129-198, the part that decides which input frame every output frame repeats.  times/lengths: n pairs, in seconds.
// Returns the output's frame count (:167-171) and fills src[0 .. that) with the input frame of each output frame, -1 for the
// frames the loops never write (they stay zero: each event swallows the frame it starts at, :178-187).  Call with src = nullptr
// to get the count only.  std::sort's order among events on the same frame is unspecified in the reference (:159, not stable);
// here the FIRST event given for a frame is the one that survives std::unique (:162-165).
int64_t oracle_freeze_plan( int64_t F, float sr, int hop, const float * times, const float * lengths, int n, int32_t * src )
	{
	using FramePair = std::array<int32_t, 2>;
	std::vector<FramePair> ev( static_cast<size_t>( std::max( n, 0 ) ) );
	for( int i = 0; i < n; ++i )
		ev[size_t( i )] = FramePair{ std::clamp( to_int( time_to_frame( times[i], sr, hop ) ), int32_t( 0 ), int32_t( F - 1 ) ),   // :152-155
		                             std::max( to_int( time_to_frame( lengths[i], sr, hop ) ), int32_t( 0 ) ) };
	std::stable_sort( ev.begin(), ev.end(), []( const FramePair & a, const FramePair & b ){ return a[0] < b[0]; } );
	ev.erase( std::unique( ev.begin(), ev.end(), []( const FramePair & a, const FramePair & b ){ return a[0] == b[0]; } ), ev.end() );
	float total = 0;                                                                 // :167-168, a float in the reference
	for( auto & e : ev ) total += float( e[1] );
	const int64_t Fo = int64_t( to_int( float( int32_t( F ) ) + total ) );            // :170-171: Frame += float
	if( !src ) return Fo;
	for( int64_t i = 0; i < Fo; ++i ) src[i] = -1;
	size_t k = 0;
	int64_t o = 0;
	for( int64_t in = 0; in < F; ++in )                                              // :176-195
		{
		if( k < ev.size() && in == ev[k][0] )
			{
			for( int32_t r = 0; r < ev[k][1]; ++r ) { if( o < Fo ) src[o] = int32_t( in ); ++o; }
			++k;
			}
		else { if( o < Fo ) src[o] = int32_t( in ); ++o; }
		}
	return Fo;
	}

// out[c][o][b] = pv[c][src[o]][b], zero where src[o] < 0: freeze's copy loops (:176-195) and cut_frames (:643-668, src[o] = start + o)
int oracle_select_frames( const float * pv_mf, int ch, int64_t F, int bins, const int32_t * src, int64_t Fo, float * out_mf )
	{
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	for( int c = 0; c < ch; ++c )
		for( int64_t o = 0; o < Fo; ++o )
			for( int b = 0; b < bins; ++b )
				out[pos( Fo, bins, c, o, b )] = ( src[o] >= 0 && src[o] < F ) ? pv[pos( F, bins, c, src[o], b )] : MF{ 0.0f, 0.0f };
	return 0;
	}

// PV.cpp:643-668: the validated ( start, count ) of cut_frames( start, end ); count 0 = a null PV
void oracle_cut_frames_range( int64_t F, int32_t start, int32_t end, int32_t * start_out, int32_t * count_out )
	{
	*start_out = 0; *count_out = 0;
	if( end <= start || F <= 0 ) return;                                             // :651
	start = std::clamp( start, int32_t( 0 ), int32_t( F - 1 ) );                      // :652-653
	end = std::clamp( end, int32_t( 0 ), int32_t( F - 1 ) );
	*start_out = start; *count_out = std::max( end - start, 0 );
	}

// PV.cpp:698-720, one input of join: its frames go to out frames [out_start, out_start + in_F) for the channels and bins both
// have; the rest of out is left as it is (join clears the output first, :706)
int oracle_place_frames( const float * in_mf, int in_ch, int64_t in_F, int in_bins, float * out_mf, int out_ch, int64_t out_F, int out_bins, int64_t out_start )
	{
	const MF * in = reinterpret_cast<const MF*>( in_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	if( out_start < 0 || out_start + in_F > out_F ) return -1;
	for( int c = 0; c < in_ch && c < out_ch; ++c )
		for( int64_t fr = 0; fr < in_F; ++fr )
			for( int b = 0; b < in_bins && b < out_bins; ++b )
				out[pos( out_F, out_bins, c, out_start + fr, b )] = in[pos( in_F, in_bins, c, fr, b )];
	return 0;
	}

// PV.cpp:362-407 harmonic_scaler.  series: float[F][H], series( { frame_to_time( frame ), harmonic } ) for harmonic 0 .. H-1 (:371-379:
// the callable sees the 0-based index).  mode 0 = add_octaves (:409-413): harmonic h (1-based) of f is f * pow( 2, h ), computed in
// double and rounded to float; mode 1 = add_harmonics (:415-419): f * ( h + 1 ) in fp32.  A harmonic frequency that converts to a
// negative bin (a NaN frequency) indexes out of bounds in the reference; here it is skipped.
int oracle_harmonic_scale( const float * pv_mf, int ch, int64_t F, int bins, float sr, const float * series, int H, int mode, float * out_mf )
	{
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	const int dft = ( bins - 1 ) * 2;
	std::memset( out, 0, sizeof( MF ) * size_t( ch ) * F * bins );                    // :369-370
	for( int c = 0; c < ch; ++c )
		for( int64_t fr = 0; fr < F; ++fr )
			for( int b = 0; b < bins; ++b )
				{
				const MF source = pv[pos( F, bins, c, fr, b )];
				if( source.f <= 1.0f ) continue;                                          // :389
				for( int h = 0; h < H; ++h )
					{
					const int hh = h + 1;                                                  // :393
					const float hf = mode == 0 ? float( double( source.f ) * std::pow( 2.0, double( hh ) ) ) : source.f * float( hh + 1 );
					const int32_t hb = to_int( frequency_to_bin( hf, sr, dft ) );          // :394
					if( hb >= bins ) break;                                                // :395
					if( hb < 0 ) continue;
					MF & dest = out[pos( F, bins, c, fr, hb )];
					const float mag = source.m * series[size_t( fr ) * H + h];             // :398
					if( dest.m < mag ) dest = MF{ mag, hf };                               // :399-400
					}
				}
	return 0;
	}

// PV/PVModify.cpp:513-605 smear_time.  The caller samples the three callables the way :520-524 and :558-560 do; this file takes:
//   smear: float[F][bins] or nullptr + smear_const (clamped to >= 0 here, :524);  gran: int32[F][bins] or nullptr + gran_const
//   (clamped to >= 1 here, :521);  dist: float[n_dist], distribution( x / dist_samples_2 ) for x in [-dist_samples_2, dist_samples_2).
// oracle_smear_time_plan: :536-564 -- the frame the output starts at, the output's frame count, and dist_samples_2.
// A NaN smear size converts to a frame count that is undefined in the reference; here such a point spreads over no frames.
static inline float smear_at( const float * smear, float smear_const, int64_t fr, int bins, int b )
	{
	const float v = smear ? smear[size_t( fr ) * bins + b] : smear_const;
	return v < 0.0f ? 0.0f : v;                                                       // std::max( s, 0.0f ), :524
	}
static inline int32_t expansion_of( float smear_c, float sr, int hop )
	{
	const int32_t e = to_int( time_to_frame( smear_c, sr, hop ) );
	return e < 0 ? 0 : e;
	}

void oracle_smear_time_plan( int64_t F, int bins, float sr, int hop, const float * smear, float smear_const, int32_t * true_left, int64_t * out_frames, int32_t * dist_samples_2 )
	{
	int64_t left = 0, right = F - 1;                                                  // :536-537
	float mx = smear_at( smear, smear_const, 0, bins, 0 );                            // FunctionSample::maximum = std::max_element
	for( int b = 0; b < bins; ++b )
		for( int64_t fr = 0; fr < F; ++fr )                                           // :538-550
			{
			const float sz = smear_at( smear, smear_const, fr, bins, b );
			const int64_t e = expansion_of( sz, sr, hop );
			left = std::min( left, fr - e );
			right = std::max( right, fr + e );
			}
	if( smear ) { for( int64_t i = 0; i < F * bins; ++i ) { const float v = smear_at( smear, 0.0f, i / bins, bins, int( i % bins ) ); if( mx < v ) mx = v; } }
	*true_left = int32_t( left );
	*out_frames = right - left;                                                       // :563
	*dist_samples_2 = expansion_of( mx, sr, hop ) * 2;                                // :555-556
	}

int oracle_smear_time( const float * pv_mf, int ch, int64_t F, int bins, float sr, int hop, const float * smear, float smear_const, const int32_t * gran, int32_t gran_const,
	const float * dist, int64_t n_dist, int32_t true_left, int64_t Fo, float * out_mf )
	{
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	for( int c = 0; c < ch; ++c )
		for( int64_t of = 0; of < Fo; ++of )
			{
			const int64_t in_frame = std::clamp<int64_t>( of + true_left, 0, F - 1 );   // :569
			for( int b = 0; b < bins; ++b )
				{
				const float smear_size_c = smear_at( smear, smear_const, in_frame, bins, b );
				const int32_t expansion = expansion_of( smear_size_c, sr, hop );          // :574
				double mag_sum = 0, freq_sum = 0, total_dist_weight = 0, dist_weight_used = 0;
				const int32_t granularity_c = std::max( gran ? gran[size_t( in_frame ) * bins + b] : gran_const, int32_t( 1 ) );   // :579, :521
				for( int64_t off = -int64_t( expansion ); off < expansion; off += granularity_c )   // :580
					{
					const float dist_input = ( float( int32_t( off ) ) / ( float( sr ) / float( hop ) ) ) / smear_size_c;   // :583 frame_to_time( offset ) / smear
					const float access_f = float( size_t( n_dist ) ) * 0.5f * ( 1 + dist_input );                           // :584
					int32_t access = to_int( access_f );
					access = std::clamp( access, int32_t( 0 ), int32_t( n_dist - 1 ) );   // :585
					const float dist_c = dist[access];
					total_dist_weight += dist_c;                                          // :587
					const int64_t source = of + true_left + off;                          // :589
					if( source < 0 || source >= F ) continue;
					const MF mf_c = pv[pos( F, bins, c, source, b )];
					dist_weight_used += dist_c;
					mag_sum += mf_c.m * dist_c;                                           // :594-595: fp32 products, fp64 sums
					freq_sum += mf_c.f * dist_c;
					}
				if( total_dist_weight > 0.0 ) mag_sum /= total_dist_weight;               // :600-601
				if( dist_weight_used > 0.0 ) freq_sum /= dist_weight_used;
				out[pos( Fo, bins, c, of, b )] = MF{ float( mag_sum ), float( freq_sum ) };
				}
			}
	return 0;
	}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// PV/PVModify.cpp:15-193  PV::modify: every input quad ( frame-1 .. frame ) x ( bin-1 .. bin ) is mapped to a quad of the output and
// rasterised there; an output point takes the loudest weighted corner offered to it.
//   mod_tf:  TF[F][bins], the callable sampled over the input's domain (:22), seconds / Hz (converted to frames / bins here, :23-26)
//   in_f:    float[ch][F][bins], mod( { frame_to_time( frame ), the MF's own frequency } ).f (:62-66)
// Arithmetic follows what g++ makes of the reference's expressions: unqualified sqrt / fabs / floor / ceil / fmin / fmax on floats are the
// C double functions there (only `using namespace std::ranges` is in scope), so the square root of the discriminant and the division
// that follows run in double and round to float once (MSVC, the reference's other toolchain, picks the float overloads: <= 1 ulp in m).
// The reference runs frames in parallel under a mutex per output frame (:71), so WHICH of several equally loud candidates gives a point
// its frequency is unspecified there; here it is the first in ( frame, bin ) order of the quads.
// A quad with a NaN or infinite corner is skipped: in the reference its bounding box is the whole output and NaNs run through every
// expression of :116-156, so that nothing is written -- except with the interpolators that ignore their argument (midpoint, floor,
// ceil), which turn a NaN position into a weight; that accident is not reproduced.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// what one output point ( x, y ) gets from one quad.  0: not inside (next y); 1: `break` (the rest of this column of the bounding box
// is skipped, :126-153); 2: a candidate { *weight, corner }
int quad_point( const float px[4], const float py[4], const float pm[4], int x, int y, int interp_kind, float * weight, int * corner )
	{
	const float D12x = px[1] - px[0], D12y = py[1] - py[0];                            // :87-90
	const float D23x = px[2] - px[1], D23y = py[2] - py[1];
	const float D34x = px[3] - px[2], D34y = py[3] - py[2];
	const float D41x = px[0] - px[3], D41y = py[0] - py[3];
	bool c = false;                                                                   // :105-109
	if( ( ( py[0] <= y && y < py[3] ) || ( py[3] <= y && y < py[0] ) ) && ( x < D41x / D41y * ( y - py[0] ) + px[0] ) ) c = !c;
	if( ( ( py[1] <= y && y < py[0] ) || ( py[0] <= y && y < py[1] ) ) && ( x < D12x / D12y * ( y - py[1] ) + px[1] ) ) c = !c;
	if( ( ( py[2] <= y && y < py[1] ) || ( py[1] <= y && y < py[2] ) ) && ( x < D23x / D23y * ( y - py[2] ) + px[2] ) ) c = !c;
	if( ( ( py[3] <= y && y < py[2] ) || ( py[2] <= y && y < py[3] ) ) && ( x < D34x / D34y * ( y - py[3] ) + px[3] ) ) c = !c;
	if( !c ) return 0;
	const float alpha[4] = { px[0], px[1] - px[0], px[3] - px[0], px[0] - px[1] + px[2] - px[3] };   // :116-117
	const float beta[4] = { py[0], py[1] - py[0], py[3] - py[0], py[0] - py[1] + py[2] - py[3] };
	const float quadA = alpha[3] * beta[2] - alpha[2] * beta[3];                      // :119-124
	const float quadB = alpha[3] * beta[0] - alpha[0] * beta[3] + alpha[1] * beta[2] - alpha[2] * beta[1] + x * beta[3] - alpha[3] * y;
	const float quadC = alpha[1] * beta[0] - alpha[0] * beta[1] + x * beta[1] - alpha[1] * y;
	float m;
	if( quadA == 0.0f )                                                               // :126-138
		{
		if( quadB == 0.0f ) return 1;
		m = -quadC / quadB;
		}
	else
		{
		const float descriminant = quadB * quadB - 4.0f * quadA * quadC;
		if( descriminant < 0 ) return 1;
		m = float( ( double( -quadB ) + std::sqrt( double( descriminant ) ) ) / double( 2.0f * quadA ) );
		}
	const float lDenominator = alpha[1] + alpha[3] * m;                               // :139-141
	if( lDenominator == 0 ) return 1;
	const float l = ( x - alpha[0] - alpha[2] * m ) / lDenominator;
	const float epsilon = 0.0001f;                                                    // :144-145
	if( std::fabs( l - 0.5f ) > 0.5f + epsilon || std::fabs( m - 0.5f ) > 0.5f + epsilon ) return 1;
	const float interpL = interpolate( interp_kind, l ), interpM = interpolate( interp_kind, m );   // :147-148
	const float w[4] = { ( 1.0f - interpL ) * ( 1.0f - interpM ) * pm[0], ( interpL ) * ( 1.0f - interpM ) * pm[1],   // :150-154
	                     ( interpL ) * ( interpM ) * pm[2], ( 1.0f - interpL ) * ( interpM ) * pm[3] };
	const float totalWeight = w[0] + w[1] + w[2] + w[3];                              // :155-156
	if( totalWeight <= 0.0f ) return 1;
	int largest = 0;                                                                  // std::max_element, :169-170
	for( int i = 1; i < 4; ++i ) if( w[largest] < w[i] ) largest = i;
	*weight = w[largest];
	*corner = largest;
	return 2;
	}
}

extern "C" {

// :28-40: the output's frame count from the sampled grid.  -2: longer than 10 minutes (the reference prints a message and returns a
// null PV, :30-34); 0: nothing to make
int64_t oracle_modify_out_frames( const float * mod_tf, int64_t F, int bins, float sr, int hop )
	{
	const TF * mod = reinterpret_cast<const TF*>( mod_tf );
	float mx = time_to_frame( mod[0].t, sr, hop );
	for( int64_t i = 1; i < F * bins; ++i ) { const float v = time_to_frame( mod[i].t, sr, hop ); if( mx < v ) mx = v; }   // ranges::max_element with projection
	const float last = std::ceil( mx );                                               // :29
	if( last / ( float( sr ) / float( hop ) ) > 60.0f * 10.0f ) return -2;            // :31
	const int32_t Fo = to_int( std::ceil( last ) );                                   // :38
	return Fo < 0 ? 0 : Fo;
	}

int oracle_modify( const float * pv_mf, int ch, int64_t F, int bins, float sr, int hop, const float * mod_tf, const float * in_f, int interp_kind, int64_t Fo, float * out_mf )
	{
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	const TF * mod = reinterpret_cast<const TF*>( mod_tf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	const int dft = ( bins - 1 ) * 2;
	std::memset( out, 0, sizeof( MF ) * size_t( ch ) * Fo * bins );                   // :40
	std::vector<float> gx( size_t( F ) * bins ), gy( size_t( F ) * bins );
	for( size_t i = 0; i < gx.size(); ++i ) { gx[i] = time_to_frame( mod[i].t, sr, hop ); gy[i] = frequency_to_bin( mod[i].f, sr, dft ); }   // :23-26
	for( int c = 0; c < ch; ++c )
		for( int64_t frame = 1; frame < F; ++frame )                                  // :70
			for( int bin = 1; bin < bins; ++bin )
				{
				const size_t idx[4] = { size_t( frame - 1 ) * bins + bin - 1, size_t( frame ) * bins + bin - 1, size_t( frame ) * bins + bin, size_t( frame - 1 ) * bins + bin };   // :76-86
				float px[4], py[4], pm[4];
				for( int k = 0; k < 4; ++k ) { px[k] = gx[idx[k]]; py[k] = gy[idx[k]]; pm[k] = pv[size_t( c ) * F * bins + idx[k]].m; }
				bool finite = true;                                                       // see the header: such quads offer nothing
				for( int k = 0; k < 4; ++k ) finite = finite && std::isfinite( px[k] ) && std::isfinite( py[k] );
				if( !finite ) continue;
				// :93-96 (fmin / fmax / floor / ceil in double on float values: exact; the conversions to Frame / Bin saturate here)
				const double minx_d = std::fmax( std::floor( std::fmin( std::fmin( px[0], px[1] ), std::fmin( px[2], px[3] ) ) ), 0 );
				const double miny_d = std::fmax( std::floor( std::fmin( std::fmin( py[0], py[1] ), std::fmin( py[2], py[3] ) ) ), 0 );
				const double maxx_d = std::fmin( std::ceil( std::fmax( std::fmax( px[0], px[1] ), std::fmax( px[2], px[3] ) ) ), double( Fo - 1 ) );
				const double maxy_d = std::fmin( std::ceil( std::fmax( std::fmax( py[0], py[1] ), std::fmax( py[2], py[3] ) ) ), double( bins - 1 ) );
				const int32_t minx = to_int( float( minx_d ) ), miny = to_int( float( miny_d ) ), maxx = to_int( float( maxx_d ) ), maxy = to_int( float( maxy_d ) );
				for( int32_t x = minx; x <= maxx; ++x )                                   // :99-101
					for( int32_t y = miny; y <= maxy; ++y )
						{
						float weight; int corner;
						const int r = quad_point( px, py, pm, x, y, interp_kind, &weight, &corner );
						if( r == 0 ) continue;
						if( r == 1 ) break;
						MF & o = out[pos( Fo, bins, c, x, y )];
						if( weight > o.m ) o = MF{ weight, in_f[size_t( c ) * F * bins + idx[corner]] };   // :172-176
						}
				}
	return 0;
	}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// PV/PVModify.cpp:387-443  PV::stretch_spline, with the cubic spline the reference vendors (spline/spline.h:284-401: natural
// boundary conditions, tridiagonal system through band_matrix's LU with its row normalisation, :187-261).  The spline restated here
// is pinned against the real header (oracle/_ref, tests/test_oracle_vs_ref.py).
//   steps: uint32[F-1], safeInterpolation( frame ) = max( uint32( interpolation( frame * frame_to_time( 1 ) ) ), 1 ) (:391-394), the
//   caller's; knot of frame k = sum of the steps before it; the output has sum( steps ) frames (:399-405).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Spline
	{
	std::vector<double> x, y, a, b, c;
	void set_points( const std::vector<double> & xs, const std::vector<double> & ys )   // spline.h:284-372, cubic branch, second_deriv = 0 both ends
		{
		x = xs; y = ys;
		const int n = int( x.size() );
		std::vector<double> lo( n, 0.0 ), di( n, 0.0 ), up( n, 0.0 ), rhs( n, 0.0 ), sd( n, 0.0 );   // A(i,i-1), A(i,i), A(i,i+1)
		for( int i = 1; i < n - 1; ++i )                                              // :302-307
			{
			lo[i] = 1.0 / 3.0 * ( x[i] - x[i - 1] );
			di[i] = 2.0 / 3.0 * ( x[i + 1] - x[i - 1] );
			up[i] = 1.0 / 3.0 * ( x[i + 1] - x[i] );
			rhs[i] = ( y[i + 1] - y[i] ) / ( x[i + 1] - x[i] ) - ( y[i] - y[i - 1] ) / ( x[i] - x[i - 1] );
			}
		di[0] = 2.0; up[0] = 0.0; rhs[0] = 0.0;                                       // :309-313
		di[n - 1] = 2.0; lo[n - 1] = 0.0; rhs[n - 1] = 0.0;                           // :323-327
		for( int i = 0; i < n; ++i )                                                  // lu_decompose, :194-204: every row scaled by 1 / a_ii
			{
			sd[i] = 1.0 / di[i];
			if( i > 0 ) lo[i] *= sd[i];
			if( i < n - 1 ) up[i] *= sd[i];
			di[i] = 1.0;
			}
		for( int k = 0; k < n - 1; ++k )                                              // :207-219
			{
			const double xk = -lo[k + 1] / di[k];
			lo[k + 1] = -xk;
			di[k + 1] = di[k + 1] + xk * up[k];
			}
		std::vector<double> yt( n );                                                  // l_solve, :222-235
		for( int i = 0; i < n; ++i )
			{
			double sum = 0;
			if( i > 0 ) sum += lo[i] * yt[i - 1];
			yt[i] = ( rhs[i] * sd[i] ) - sum;
			}
		b.assign( n, 0.0 );                                                           // r_solve, :237-250
		for( int i = n - 1; i >= 0; --i )
			{
			double sum = 0;
			if( i < n - 1 ) sum += up[i] * b[i + 1];
			b[i] = ( yt[i] - sum ) / di[i];
			}
		a.assign( n, 0.0 ); c.assign( n, 0.0 );                                       // :343-349
		for( int i = 0; i < n - 1; ++i )
			{
			a[i] = 1.0 / 3.0 * ( b[i + 1] - b[i] ) / ( x[i + 1] - x[i] );
			c[i] = ( y[i + 1] - y[i] ) / ( x[i + 1] - x[i] ) - 1.0 / 3.0 * ( 2.0 * b[i] + b[i + 1] ) * ( x[i + 1] - x[i] );
			}
		const double h = x[n - 1] - x[n - 2];                                         // :366-370
		a[n - 1] = 0.0;
		c[n - 1] = 3.0 * a[n - 2] * h * h + 2.0 * b[n - 2] * h + c[n - 2];
		}
	double operator()( double t ) const                                               // :375-397
		{
		const size_t n = x.size();
		const int idx = std::max( int( std::lower_bound( x.begin(), x.end(), t ) - x.begin() ) - 1, 0 );
		const double h = t - x[idx];
		if( t < x[0] ) return ( b[0] * h + c[0] ) * h + y[0];
		if( t > x[n - 1] ) return ( b[n - 1] * h + c[n - 1] ) * h + y[n - 1];
		return ( ( a[idx] * h + b[idx] ) * h + c[idx] ) * h + y[idx];
		}
	};
}

extern "C" {

void oracle_spline( const double * x, const double * y, int n, const double * t, int nt, double * out )
	{
	Spline s;
	s.set_points( std::vector<double>( x, x + n ), std::vector<double>( y, y + n ) );
	for( int i = 0; i < nt; ++i ) out[i] = s( t[i] );
	}

int64_t oracle_stretch_spline_out_frames( const uint32_t * steps, int64_t F )
	{
	int32_t total = 0;                                                                // :399-405: Frame += uint32
	for( int64_t i = 0; i + 1 < F; ++i ) total = int32_t( uint32_t( total ) + steps[i] );
	return total;
	}

int oracle_stretch_spline( const float * pv_mf, int ch, int64_t F, int bins, const uint32_t * steps, int64_t Fo, float * out_mf )
	{
	if( F < 3 ) return -1;                                                            // spline.h:288 asserts more than two points
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::vector<double> Xs( static_cast<size_t>( F ) ), ms( static_cast<size_t>( F ) ), fs( static_cast<size_t>( F ) );
	int32_t run = 0;
	for( int64_t fr = 0; fr + 1 < F; ++fr ) { Xs[size_t( fr )] = run; run = int32_t( uint32_t( run ) + steps[fr] ); }   // :400-405
	Xs[size_t( F - 1 )] = run;
	for( int c = 0; c < ch; ++c )
		for( int b = 0; b < bins; ++b )                                               // :413-441
			{
			for( int64_t fr = 0; fr < F; ++fr ) { ms[size_t( fr )] = pv[pos( F, bins, c, fr, b )].m; fs[size_t( fr )] = pv[pos( F, bins, c, fr, b )].f; }
			Spline sm, sf;
			sm.set_points( Xs, ms );
			sf.set_points( Xs, fs );
			for( int64_t fr = 0; fr < Fo; ++fr ) out[pos( Fo, bins, c, fr, b )] = MF{ float( sm( double( fr ) ) ), float( sf( double( fr ) ) ) };
			}
	return 0;
	}

} // extern "C"
