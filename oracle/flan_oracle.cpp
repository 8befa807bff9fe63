// flan_oracle.cpp -- CPU restatement of Flan's phase-vocoder hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under flan_amd/ or include/ may link, import or call this file.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
//
// What it is: a from-scratch, scalar restatement of the reference algorithm with the SAME rounding
// sequence (fp32 where the reference computes in fp32, fp64 where it computes in fp64, the same libm
// entry points: atan2f / hypotf / roundf / sinf / cosf / fmod).  Every function cites the reference
// file:line it follows (paths relative to /root/reference/src/flan).
//
// Pinning status (see DESIGN.md "Oracle"):
//   * per-bin math, Hann window, PVBuffer unit conversions: checked BIT-EXACT against the reference's own
//     translation units (phase_vocoder.cpp, WindowFunctions.cpp, PV/PVBuffer.cpp) compiled unmodified into
//     oracle/_ref/libflanref.so (tests/test_oracle_vs_ref.py).
//   * frame loops (Conversions/AudioPV.cpp), frame processors (PV/PVModify.cpp, PV/PV.cpp): those TUs are
//     UNBUILDABLE here (need FFTW3f, libsndfile, MSVC-only std::_Pi) and the reference ships no tests or
//     golden vectors, so they are pinned only by the known-answer anchors recorded in SURVEY.md section 8c
//     (tests/test_oracle_anchors.py).  PARITY UNPINNED by reference fixtures for those loops.
//   * FFT: the reference calls FFTW3f (external, version unpinned, absent).  r2c/c2r are restated from their
//     published definition (unnormalised; c2r ignores Im of DC and Nyquist), evaluated in fp64 and rounded
//     to fp32 once -- the central estimate of any correct fp32 FFT.
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

namespace {

struct MF { float m, f; };

// defines.h:44-45  (pi is a float: acos(-1.0f); pi2 = pi * 2.0f)
const float k_pi  = std::acos( -1.0f );
const float k_pi2 = k_pi * 2.0f;

// ---------------------------------------------------------------------------------------------------------
// FFT (stands in for FFTW3f at FFTHelper.cpp:21-24,41,47), fp64.  Powers of two: iterative radix-2, twiddles from
// sincos in fp64.  Every other size (FFTW takes any: FFTHelper.cpp:16-26 passes the caller's dft_size through):
// recursive mixed radix over the prime factors (a prime factor p costs p^2 per butterfly), twiddles from an exact
// table of the unit circle worked out in long double.  tests/test_oracle_fft.py holds both against numpy.fft.
// ---------------------------------------------------------------------------------------------------------
struct FFTPlan
	{
	int n = 0;
	bool pow2 = true;
	std::vector<std::complex<double>> w; // w[k] = exp(-2 pi i k / n), k < n/2 (power of two) or k < n (any size)
	std::vector<int> rev;
	explicit FFTPlan( int n_ ) : n( n_ ), pow2( n_ > 0 && ( n_ & ( n_ - 1 ) ) == 0 )
		{
		if( !pow2 )
			{
			w.resize( n );
			const long double two_pi = 6.283185307179586476925286766559005768L;
			for( int k = 0; k < n; ++k ) w[k] = std::complex<double>( double( cosl( two_pi * k / n ) ), double( -sinl( two_pi * k / n ) ) );
			w[0] = std::complex<double>( 1.0, 0.0 );
			if( n % 2 == 0 ) w[n / 2] = std::complex<double>( -1.0, 0.0 );
			if( n % 4 == 0 ) { w[n / 4] = std::complex<double>( 0.0, -1.0 ); w[3 * ( n / 4 )] = std::complex<double>( 0.0, 1.0 ); }
			return;
			}
		w.resize( n / 2 ); rev.resize( n );
		const double pi = 3.14159265358979323846;
		for( int k = 0; k < n / 2; ++k )
			w[k] = std::complex<double>( std::cos( -2.0 * pi * k / n ), std::sin( -2.0 * pi * k / n ) );
		int bits = 0; while( ( 1 << bits ) < n ) ++bits;
		for( int i = 0; i < n; ++i )
			{
			int r = 0;
			for( int b = 0; b < bits; ++b ) if( i & ( 1 << b ) ) r |= 1 << ( bits - 1 - b );
			rev[i] = r;
			}
		}
	// out[k] = sum_j in[j * stride] exp( sign 2 pi i j k / m ), m a divisor of n: decimation in time over m's smallest prime factor
	void any_rec( const std::complex<double> * in, int stride, int m, std::complex<double> * out, int sign ) const
		{
		if( m == 1 ) { out[0] = in[0]; return; }
		int p = 2;
		while( m % p ) ++p;
		const int q = m / p, root = n / m;                                   // exp( -2 pi i j / m ) = w[j * root]
		std::vector<std::complex<double>> sub( m );
		for( int r = 0; r < p; ++r ) any_rec( in + size_t( r ) * stride, stride * p, q, sub.data() + size_t( r ) * q, sign );
		for( int k = 0; k < q; ++k )
			for( int c = 0; c < p; ++c )
				{
				const int kk = k + c * q;                                     // output index; input r contributes sub_r[k] w_m^( r kk )
				std::complex<double> acc = sub[k];
				for( int r = 1; r < p; ++r )
					{
					std::complex<double> t = w[size_t( ( int64_t( r ) * kk ) % m ) * root];
					if( sign > 0 ) t = std::conj( t );
					acc += sub[size_t( r ) * q + k] * t;
					}
				out[kk] = acc;
				}
		}
	// in-place complex FFT; sign=-1 forward, +1 inverse (unnormalised)
	void run( std::complex<double> * a, int sign ) const
		{
		if( !pow2 )
			{
			std::vector<std::complex<double>> out( n );
			any_rec( a, 1, n, out.data(), sign );
			std::copy( out.begin(), out.end(), a );
			return;
			}
		for( int i = 0; i < n; ++i ) if( i < rev[i] ) std::swap( a[i], a[rev[i]] );
		for( int len = 2; len <= n; len <<= 1 )
			{
			const int half = len >> 1, step = n / len;
			for( int i = 0; i < n; i += len )
				for( int j = 0; j < half; ++j )
					{
					std::complex<double> t = w[j * step];
					if( sign > 0 ) t = std::conj( t );
					const std::complex<double> u = a[i + j], v = a[i + j + half] * t;
					a[i + j] = u + v;
					a[i + j + half] = u - v;
					}
			}
		}
	};

bool is_pow2( int n ) { return n > 0 && ( n & ( n - 1 ) ) == 0; }

// r2c: X[k] = sum_n x[n] exp(-2 pi i k n / N), k = 0..N/2   (fftwf_plan_dft_r2c_1d semantics)
void r2c( const FFTPlan & p, const float * x, std::complex<float> * X, std::vector<std::complex<double>> & tmp )
	{
	for( int i = 0; i < p.n; ++i ) tmp[i] = std::complex<double>( x[i], 0.0 );
	p.run( tmp.data(), -1 );
	for( int k = 0; k <= p.n / 2; ++k ) X[k] = std::complex<float>( float( tmp[k].real() ), float( tmp[k].imag() ) );
	// a real-input transform (FFTW's r2c) computes X[0] and X[N/2] as real numbers: their imaginary parts are exact zeros, not the
	// 1e-17 a complex mixed-radix pass leaves there (the radix-2 path yields the zeros by itself)
	if( !p.pow2 ) { X[0] = std::complex<float>( X[0].real(), 0.0f ); if( p.n % 2 == 0 ) X[p.n / 2] = std::complex<float>( X[p.n / 2].real(), 0.0f ); }
	}

// c2r: x[n] = sum_{k=0}^{N-1} X[k] exp(+2 pi i k n / N) with X[N-k] = conj X[k]; Im X[0], Im X[N/2] ignored;
// unnormalised (fftwf_plan_dft_c2r_1d semantics).
void c2r( const FFTPlan & p, const std::complex<float> * X, float * x, std::vector<std::complex<double>> & tmp )
	{
	const int n = p.n;
	tmp[0] = std::complex<double>( X[0].real(), 0.0 );
	tmp[n / 2] = std::complex<double>( X[n / 2].real(), 0.0 );
	for( int k = 1; k < n / 2; ++k )
		{
		tmp[k] = std::complex<double>( X[k].real(), X[k].imag() );
		tmp[n - k] = std::conj( tmp[k] );
		}
	p.run( tmp.data(), +1 );
	for( int i = 0; i < n; ++i ) x[i] = float( tmp[i].real() );
	}

// WindowFunctions.cpp:8-13.  Under libstdc++ the unqualified cos(float) call resolves to ::cos(double):
// the float product 2.0f*pi*x is widened, the cosine is taken in double, and 1.0f - (double) is a double
// subtraction; 0.5f * (double) is a double product, narrowed to float on return.
float hann( float x )
	{
	return float( 0.5f * ( 1.0f - std::cos( double( 2.0f * k_pi * x ) ) ) );
	}

// PVBuffer.cpp:443-446  bin_to_frequency(b) = b * float(sr) / float(dft)
float bin_to_frequency( float b, float sr, int dft ) { return b * float( sr ) / float( dft ); }
// PVBuffer.cpp:438-441
float frequency_to_bin( float f, float sr, int dft ) { return f / ( float( sr ) / float( dft ) ); }
// PVBuffer.cpp:428-431 / 433-436
float time_to_frame( float t, float sr, int hop ) { return t * float( sr ) / float( hop ); }
float frame_to_time( float f, float sr, int hop ) { return f / ( float( sr ) / float( hop ) ); }

// phase_vocoder.cpp:5-53
MF phase_vocoder( double & phase_buffer, std::complex<float> cpx, float bin_frequency, float analysis_rate, float sample_rate )
	{
	const bool use_wrapping = analysis_rate < sample_rate;
	const float phase = std::atan2( cpx.imag(), cpx.real() );          // std::arg(complex<float>) == atan2f
	const float phase_diff = float( phase - phase_buffer );             // double subtraction, narrowed (:44)
	phase_buffer = phase;                                               // :45
	const float expected_phase_diff = bin_frequency / analysis_rate * k_pi2; // :47
	const float delta_phase = phase_diff - expected_phase_diff;        // :48
	const float wrapped = use_wrapping ? ( delta_phase - k_pi2 * std::round( delta_phase / k_pi2 ) ) : delta_phase; // :39-42,49
	const float delta_frequency = wrapped * analysis_rate / k_pi2;     // :50
	return { std::hypot( cpx.real(), cpx.imag() ), bin_frequency + delta_frequency }; // std::abs(complex<float>) == hypotf
	}

// phase_vocoder.cpp:55-61
std::complex<float> inverse_phase_vocoder( double & phase_buffer, MF mf, float analysis_rate )
	{
	const float phase_diff = mf.f / analysis_rate * k_pi2;
	phase_buffer += phase_diff;
	if( phase_buffer > k_pi2 ) phase_buffer = std::fmod( phase_buffer, double( k_pi2 ) );
	const float th = float( phase_buffer );
	return std::complex<float>( mf.m * std::cos( th ), mf.m * std::sin( th ) ); // std::polar(rho, theta)
	}

} // namespace

extern "C" {

float oracle_pi2() { return k_pi2; }

void oracle_hann_window( float * w, int window_size )
	{
	// AudioPV.cpp:30-34
	for( int i = 0; i < window_size; ++i ) w[i] = hann( float( i ) / float( window_size - 1 ) );
	}

float oracle_hann( float x ) { return hann( x ); }

void oracle_phase_vocoder( double * phase_buffer, float re, float im, float bin_frequency, float analysis_rate, float sample_rate, float * m, float * f )
	{
	const MF r = phase_vocoder( *phase_buffer, std::complex<float>( re, im ), bin_frequency, analysis_rate, sample_rate );
	*m = r.m; *f = r.f;
	}

void oracle_inverse_phase_vocoder( double * phase_buffer, float m, float f, float analysis_rate, float * re, float * im )
	{
	const std::complex<float> c = inverse_phase_vocoder( *phase_buffer, MF{ m, f }, analysis_rate );
	*re = c.real(); *im = c.imag();
	}

float oracle_bin_to_frequency( float b, float sr, int dft ) { return bin_to_frequency( b, sr, dft ); }
float oracle_frequency_to_bin( float f, float sr, int dft ) { return frequency_to_bin( f, sr, dft ); }
float oracle_time_to_frame( float t, float sr, int hop ) { return time_to_frame( t, sr, hop ); }
float oracle_frame_to_time( float f, float sr, int hop ) { return frame_to_time( f, sr, hop ); }

// r2c / c2r exposed for FFT unit tests
int oracle_r2c( const float * x, int n, float * X /* (n/2+1)*2 */ )
	{
	if( n < 2 || n % 2 ) return -1;
	FFTPlan p( n ); std::vector<std::complex<double>> tmp( n );
	r2c( p, x, reinterpret_cast<std::complex<float>*>( X ), tmp );
	return 0;
	}
int oracle_c2r( const float * X, int n, float * x )
	{
	if( n < 2 || n % 2 ) return -1;
	FFTPlan p( n ); std::vector<std::complex<double>> tmp( n );
	c2r( p, reinterpret_cast<const std::complex<float>*>( X ), x, tmp );
	return 0;
	}

// `reps` forward + inverse transforms of one buffer with ONE plan, as the frame loops below use it (bench.py: the FFT's share of the
// checker's time -- this FFT is not FFTW).  Returns a value that depends on the result so the loop cannot be dropped.
double oracle_fft_pairs( int n, int reps )
	{
	if( !is_pow2( n ) ) return -1.0;
	FFTPlan p( n ); std::vector<std::complex<double>> tmp( n );
	std::vector<float> x( n ); std::vector<std::complex<float>> X( n / 2 + 1 );
	for( int i = 0; i < n; ++i ) x[i] = float( ( i * 2654435761u ) >> 8 ) * ( 1.0f / 16777216.0f ) - 0.5f;
	double acc = 0.0;
	for( int r = 0; r < reps; ++r )
		{
		r2c( p, x.data(), X.data(), tmp );
		c2r( p, X.data(), x.data(), tmp );
		const float g = 1.0f / float( n );
		for( int i = 0; i < n; ++i ) x[i] *= g;
		acc += x[r % n];
		}
	return acc;
	}

// AudioPV.cpp:17  numHops = ceil( num_frames / hop ) + 1 with INTEGER division
int64_t oracle_num_pv_frames( int64_t num_audio_frames, int hop ) { return num_audio_frames / hop + 1; }

// Audio::convert_to_PV, AudioPV.cpp:12-78.  audio: [ch][n] channel-major (AudioBuffer.cpp:479-482);
// out: MF[ch][F][bins] (PVBuffer.cpp:526-529), F = n/hop + 1, bins = dft/2+1.
int oracle_analyze( const float * audio, int num_channels, int64_t n, float sample_rate, int window_size, int hop, int dft_size, float * out_mf )
	{
	if( dft_size < 4 || dft_size % 2 || window_size > dft_size || window_size < 2 || hop < 1 ) return -1;   // any even size, like FFTW behind FFTHelper.cpp:16-26
	const int num_bins = dft_size / 2 + 1;                       // :15
	const int64_t num_hops = n / hop + 1;                        // :17
	const float analysis_rate = sample_rate / hop;               // :26
	MF * out = reinterpret_cast<MF*>( out_mf );

	std::vector<float> hann_window( window_size );              // :30-34
	oracle_hann_window( hann_window.data(), window_size );

	std::vector<double> phase_buffer( num_bins );               // :37
	FFTPlan plan( dft_size );                                    // :38
	std::vector<float> real( dft_size );
	std::vector<std::complex<float>> cpx( num_bins );
	std::vector<std::complex<double>> tmp( dft_size );

	for( int channel = 0; channel < num_channels; ++channel )   // :41
		{
		std::fill( phase_buffer.begin(), phase_buffer.end(), 0.0 ); // :44
		const float * x = audio + int64_t( channel ) * n;
		for( int64_t frame = 0; frame < num_hops; ++frame )     // :47
			{
			const int64_t start = int64_t( hop ) * frame - window_size / 2; // :52
			for( int i = 0; i < window_size; ++i )               // :54-62
				{
				const int64_t s = start + i;
				const float v = ( s < 0 || n <= s ) ? 0.0f : x[s];
				real[i] = v * hann_window[i];
				}
			std::fill( real.begin() + window_size, real.end(), 0.0f ); // :65
			r2c( plan, real.data(), cpx.data(), tmp );           // :67
			MF * row = out + ( int64_t( channel ) * num_hops + frame ) * num_bins;
			for( int bin = 0; bin < num_bins; ++bin )            // :69-73
				row[bin] = phase_vocoder( phase_buffer[bin], cpx[bin], bin_to_frequency( float( bin ), sample_rate, dft_size ), analysis_rate, sample_rate );
			}
		}
	return 0;
	}

// PVBuffer.cpp:381-384  get_hop_size = Frame( sample_rate / analysis_rate )
int oracle_hop_size( float sample_rate, float analysis_rate ) { return int( sample_rate / analysis_rate ); }

// PV::convert_to_audio, AudioPV.cpp:86-139.  pv: MF[ch][F][bins]; out: float[ch][F*hop].
// Returns 1 if the buffer held a NaN/Inf (reference prints a warning and continues, :88-89), 0 otherwise, <0 on error.
int oracle_synthesize( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate, float analysis_rate, int window_size, float * out )
	{
	const int dft_size = ( num_bins - 1 ) * 2;                   // PVBuffer.cpp:356-359
	const int hop = int( sample_rate / analysis_rate );          // PVBuffer.cpp:381-384
	if( dft_size < 4 || dft_size % 2 || hop < 1 || window_size > dft_size ) return -1;
	const MF * pv = reinterpret_cast<const MF*>( pv_mf );
	const int64_t out_frames = num_frames * hop;                 // :93

	int nan_flag = 0;                                            // :88 / PVBuffer.cpp:44-50
	for( int64_t i = 0; i < int64_t( num_channels ) * num_frames * num_bins; ++i )
		if( std::isnan( pv[i].m ) || std::isnan( pv[i].f ) || std::isinf( pv[i].m ) || std::isinf( pv[i].f ) ) { nan_flag = 1; break; }

	std::fill( out, out + int64_t( num_channels ) * out_frames, 0.0f ); // Audio( format ) zero-filled

	std::vector<float> hann_window( window_size );
	const float window_scale = 2.67f / ( dft_size * window_size / hop ); // :99 (integer arithmetic in the divisor)
	for( int i = 0; i < window_size; ++i )                        // :100-103
		hann_window[i] = hann( float( i ) / float( window_size - 1 ) ) * window_scale;

	std::vector<double> phase_buffer( num_bins );                // :105
	FFTPlan plan( dft_size );
	std::vector<float> real( dft_size );
	std::vector<std::complex<float>> cpx( num_bins );
	std::vector<std::complex<double>> tmp( dft_size );

	for( int channel = 0; channel < num_channels; ++channel )    // :108
		{
		std::fill( phase_buffer.begin(), phase_buffer.end(), 0.0 ); // :111
		float * o = out + int64_t( channel ) * out_frames;
		for( int64_t frame = 0; frame < num_frames; ++frame )    // :113
			{
			const MF * row = pv + ( int64_t( channel ) * num_frames + frame ) * num_bins;
			for( int bin = 0; bin < num_bins; ++bin )             // :117-120
				cpx[bin] = inverse_phase_vocoder( phase_buffer[bin], row[bin], analysis_rate );
			c2r( plan, cpx.data(), real.data(), tmp );            // :122
			const int64_t start = int64_t( hop ) * frame - window_size / 2; // :125
			const int64_t end = start + window_size;
			const int64_t start_b = std::max<int64_t>( start, 0 );
			const int64_t end_b = std::min<int64_t>( end, out_frames );
			for( int64_t i = start_b - start; i < end_b - start; ++i ) // :133-134
				o[start + i] += real[i] * hann_window[i];
			}
		}
	return nan_flag;
	}

// ---------------------------------------------------------------------------------------------------------
// Frame processors
// ---------------------------------------------------------------------------------------------------------

// modify_time_base, PVModify.cpp:307-362, linear interpolator (Interpolator.cpp:50-56).
// pv: MF[ch][F][bins]; mod: seconds, float[F][bins] (FunctionSample2d, FunctionSample.h:173-199).
// out_frames must equal oracle_modify_time_out_frames(); out: MF[ch][out_frames][bins].
int64_t oracle_modify_time_out_frames( const float * mod, int64_t num_frames, int num_bins, float sample_rate, int hop )
	{
	// :312  ceil( time_to_frame( mod.maximum() ) ), then narrowed float -> Frame (:315)
	float mx = mod[0];
	for( int64_t i = 1; i < num_frames * num_bins; ++i ) mx = std::max( mx, mod[i] );
	const float last = std::ceil( time_to_frame( mx, sample_rate, hop ) );
	return int64_t( int32_t( last ) );
	}

float oracle_interpolate( int kind, float x );    // processors_oracle.cpp (Utility/Interpolator.cpp:14-101, pinned against the reference TU)

int oracle_modify_time_interp( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate, int hop,
	const float * mod, int64_t out_frames, int interp, float * out_mf );
int oracle_modify_time( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate, int hop,
	const float * mod, int64_t out_frames, float * out_mf )
	{
	return oracle_modify_time_interp( pv_mf, num_channels, num_frames, num_bins, sample_rate, hop, mod, out_frames, 0, out_mf );
	}

int oracle_modify_time_interp( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate, int hop,
	const float * mod, int64_t out_frames, int interp, float * out_mf )
	{
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::memset( out, 0, sizeof( MF ) * size_t( num_channels ) * out_frames * num_bins ); // clear_buffer, :317
	for( int channel = 0; channel < num_channels; ++channel )     // :319
		for( int bin = 0; bin < num_bins; ++bin )                  // :325
			for( int64_t frame = 1; frame < num_frames; ++frame )  // :328
				{
				const float lFrame = time_to_frame( mod[( frame - 1 ) * num_bins + bin], sample_rate, hop ); // :330
				const float rFrame = time_to_frame( mod[frame * num_bins + bin], sample_rate, hop );         // :331
				const bool forward = rFrame > lFrame;              // :332
				const int32_t start_frame = int32_t( forward ? std::ceil( lFrame ) : std::floor( lFrame ) ); // :334
				const int32_t end_frame   = int32_t( forward ? std::ceil( rFrame ) : std::floor( rFrame ) ); // :335
				const MF lMF = in[( int64_t( channel ) * num_frames + frame - 1 ) * num_bins + bin];
				const MF rMF = in[( int64_t( channel ) * num_frames + frame ) * num_bins + bin];
				bool stop_bin = false;
				for( int32_t x = start_frame; x != end_frame; forward ? ++x : --x ) // :340
					{
					if( x < 0 || out_frames <= x ) continue;       // :342
					const float mix = oracle_interpolate( interp, ( x - lFrame ) / ( rFrame - lFrame ) ); // :344 interp( ... ); linear = identity
					const float w0 = ( 1.0f - mix ) * lMF.m;
					const float w1 = mix * rMF.m;
					const float totalWeight = w0 + w1;
					const float weightedFreqSum = w0 * lMF.f + w1 * rMF.f;
					if( totalWeight == 0.0f ) { stop_bin = true; break; } // :350-351: `return` from the per-frame lambda => next frame
					MF & o = out[( int64_t( channel ) * out_frames + x ) * num_bins + bin];
					o.f = ( o.f * o.m + weightedFreqSum ) / ( o.m + totalWeight ); // :354
					o.m += totalWeight;                             // :355
					}
				(void) stop_bin; // the lambda returns for this (frame) only; the frame loop continues
				}
	return 0;
	}

// PV::stretch front half, PVModify.cpp:371-382: sample factor on the (frame,bin) grid, inclusive prefix-sum over
// frames per bin (in place, fp32), then frame_to_time.  factor: float[F][bins] on entry, seconds on exit.
void oracle_stretch_map( float * factor, int64_t num_frames, int num_bins, float sample_rate, int hop )
	{
	for( int bin = 0; bin < num_bins; ++bin )                      // :376-378
		for( int64_t frame = 1; frame < num_frames; ++frame )
			factor[frame * num_bins + bin] += factor[( frame - 1 ) * num_bins + bin];
	for( int64_t i = 0; i < num_frames * num_bins; ++i )          // :381-382
		factor[i] = frame_to_time( factor[i], sample_rate, hop );
	}

// modify_frequency_base, PVModify.cpp:196-257, linear interpolator.
// mod: float[F][bins] Hz (where each grid bin centre maps); in_modified: float[ch][F][bins] Hz (new frequency of each MF).
int oracle_modify_frequency_interp( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate,
	const float * mod, const float * in_modified, int interp, float * out_mf );
int oracle_modify_frequency( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate,
	const float * mod, const float * in_modified, float * out_mf )
	{
	return oracle_modify_frequency_interp( pv_mf, num_channels, num_frames, num_bins, sample_rate, mod, in_modified, 0, out_mf );
	}

int oracle_modify_frequency_interp( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate,
	const float * mod, const float * in_modified, int interp, float * out_mf )
	{
	const int dft = ( num_bins - 1 ) * 2;
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::memset( out, 0, sizeof( MF ) * size_t( num_channels ) * num_frames * num_bins ); // :205
	for( int channel = 0; channel < num_channels; ++channel )     // :207
		{
		const float * inmod = in_modified + int64_t( channel ) * num_frames * num_bins; // :209
		for( int64_t frame = 0; frame < num_frames; ++frame )     // :211
			for( int bin = 1; bin < num_bins; ++bin )              // :214
				{
				const int64_t hi = frame * num_bins + bin;          // :217
				const float loBin = frequency_to_bin( mod[hi - 1], sample_rate, dft ); // :218
				const float hiBin = frequency_to_bin( mod[hi], sample_rate, dft );     // :219
				const bool forward = hiBin > loBin;                 // :220
				const int32_t loR = int32_t( forward ? std::ceil( loBin ) : std::floor( loBin ) ); // :222
				const int32_t hiR = int32_t( forward ? std::ceil( hiBin ) : std::floor( hiBin ) ); // :223
				const int32_t start_bin = std::clamp( loR, 0, num_bins - 1 ); // :224
				const int32_t end_bin   = std::clamp( hiR, 0, num_bins - 1 ); // :225
				const MF * row = in + ( int64_t( channel ) * num_frames + frame ) * num_bins;
				const MF loMF{ row[bin - 1].m, inmod[frame * num_bins + bin - 1] }; // :227
				const MF hiMF{ row[bin].m,     inmod[frame * num_bins + bin] };     // :228
				MF * orow = out + ( int64_t( channel ) * num_frames + frame ) * num_bins;
				for( int32_t y = start_bin; y != end_bin; forward ? ++y : --y ) // :230
					{
					const float mix = oracle_interpolate( interp, ( float( y ) - loBin ) / ( hiBin - loBin ) ); // :232 interp( ... ); linear = identity
					const float w0 = ( 1.0f - mix ) * loMF.m;
					const float w1 = mix * hiMF.m;
					const MF & mx = w0 < w1 ? loMF : hiMF;          // :237
					MF & o = orow[y];
					if( mx.m > o.m )                                // :239
						{
						o.m += mx.m;                                 // :241
						o.f = mx.f;                                  // :242
						}
					}
				}
		}
	return 0;
	}

// PV::repitch front half, PVModify.cpp:273-302.  factor: float[F][bins] on entry -> Hz map on exit;
// in_modified: float[ch][F][bins] out.
void oracle_repitch_map( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate,
	float * factor, float * in_modified )
	{
	const int dft = ( num_bins - 1 ) * 2;
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	for( int64_t frame = 0; frame < num_frames; ++frame )          // :278-280
		for( int bin = 1; bin < num_bins; ++bin )
			factor[frame * num_bins + bin] += factor[frame * num_bins + bin - 1];
	for( int64_t i = 0; i < num_frames * num_bins; ++i )          // :283-284
		factor[i] = bin_to_frequency( factor[i], sample_rate, dft );
	for( int channel = 0; channel < num_channels; ++channel )     // :289-302
		for( int64_t frame = 0; frame < num_frames; ++frame )
			for( int bin = 0; bin < num_bins; ++bin )
				{
				const int64_t idx = ( int64_t( channel ) * num_frames + frame ) * num_bins + bin;
				const float fbin = std::clamp( frequency_to_bin( in[idx].f, sample_rate, dft ), 0.0f, float( num_bins - 1 ) - 0.0001f );
				const int32_t lo = int32_t( std::floor( fbin ) );
				const int32_t hi = lo + 1;
				const float lo_freq = factor[frame * num_bins + lo];
				const float hi_freq = factor[frame * num_bins + hi];
				const float r = fbin - lo;
				in_modified[idx] = lo_freq * ( 1.0f - r ) + hi_freq * r;
				}
	}

// PV::shape, PV.cpp:421-458, for the affine family  shaped = { a*m + b, c*f + d }  (what a device kernel can run without
// calling back into host code), with and without shift alignment.
int oracle_shape_affine( const float * pv_mf, int num_channels, int64_t num_frames, int num_bins, float sample_rate,
	float a, float b, float c, float d, int use_shift_alignment, float * out_mf )
	{
	const int dft = ( num_bins - 1 ) * 2;
	const MF * in = reinterpret_cast<const MF*>( pv_mf );
	MF * out = reinterpret_cast<MF*>( out_mf );
	std::memset( out, 0, sizeof( MF ) * size_t( num_channels ) * num_frames * num_bins ); // :426
	for( int channel = 0; channel < num_channels; ++channel )
		for( int64_t frame = 0; frame < num_frames; ++frame )
			{
			const MF * row = in + ( int64_t( channel ) * num_frames + frame ) * num_bins;
			MF * orow = out + ( int64_t( channel ) * num_frames + frame ) * num_bins;
			for( int bin = 0; bin < num_bins; ++bin )              // :433
				{
				const MF inMF = row[bin];
				const MF shaped{ a * inMF.m + b, c * inMF.f + d };  // shaper( inMF ), :436
				if( use_shift_alignment )                           // :438-448
					{
					const int32_t binShift = int32_t( bin - frequency_to_bin( inMF.f, sample_rate, dft ) );            // :440 (float -> Bin)
					const int32_t shapedBin = int32_t( frequency_to_bin( shaped.f, sample_rate, dft ) + binShift );    // :441 (float sum -> Bin)
					if( shapedBin < 0 || num_bins <= shapedBin ) continue;
					MF & o = orow[shapedBin];
					if( shaped.m > o.m ) o = shaped;
					}
				else orow[bin] = shaped;                            // :452
				}
			}
	return 0;
	}

// Audio::convert_to_mid_side, AudioConversions.cpp:32-51 (stereo only; also convert_to_left_right :53-56)
int oracle_mid_side( const float * in, int64_t n, float * out )
	{
	const float sqrt2 = std::sqrt( 2.0f );
	for( int64_t i = 0; i < n; ++i )
		{
		out[i]     = ( in[i] + in[n + i] ) / sqrt2;
		out[n + i] = ( in[i] - in[n + i] ) / sqrt2;
		}
	return 0;
	}

// Counter-based synthetic noise shared by CPU and GPU: uniform [-1,1) from a 32-bit hash of
// (seed, channel, n).  Defined by this project, not by the reference.
static inline uint32_t hash32( uint32_t x )
	{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
	}
void oracle_noise( float * out, int num_channels, int64_t n, uint32_t seed )
	{
	for( int c = 0; c < num_channels; ++c )
		for( int64_t i = 0; i < n; ++i )
			{
			const uint32_t u = hash32( hash32( seed ^ ( uint32_t( c ) * 0x9E3779B9U ) ) + uint32_t( i ) * 0x85EBCA6BU + uint32_t( uint64_t( i ) >> 32 ) );
			out[int64_t( c ) * n + i] = float( u >> 8 ) * ( 1.0f / 8388608.0f ) - 1.0f;
			}
	}

} // extern "C"
