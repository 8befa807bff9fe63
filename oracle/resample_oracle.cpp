// resample_oracle.cpp -- CPU restatement of Audio::resample for the 2:1 decimation case (BASELINE config 5: 96 kHz -> 48 kHz).
//
// TEST INFRASTRUCTURE ONLY (same rules as flan_oracle.cpp).  Pinned against the REAL r8brain resampler the reference vendors
// (oracle/_ref/libr8bref.so, tests/test_oracle_resample.py).
//
// What the reference does (Audio/AudioConversions.cpp:14-30): r8b::CDSPResampler( src, dst, num_frames ) with default
// parameters -- transition band 2 %, stop-band attenuation 206.91 dB, linear phase (r8brain/CDSPResampler.h:115-118) -- and ONE
// oneshot<float,float> over the whole channel-major buffer, i.e. all channels as one stream (filter ringing bleeds across
// channel boundaries; reproduced).  For src = 2 dst the constructor takes the {1,2} "common ratio" branch
// (CDSPResampler.h:144-170): a single CDSPBlockConvolver with the low-pass of CDSPFIRFilter::buildLPFilter
// (CDSPFIRFilter.h:227-493) at normalised cut-off 1/2, gain 1, followed by 2:1 decimation with the filter latency
// (fl2 samples) consumed (CDSPBlockConvolver.h:62-184).  The block convolver is an FFT overlap-save implementation of plain
// convolution; restated here as the direct sum in fp64:
//        out[k] = float( sum_{j=-fl2..fl2} h[j] * x[2k - j] ),   x = 0 outside the buffer.
// The low-pass is a Kaiser(beta 125)-power windowed sinc whose length / cut-off / window power come from r8brain's fitted
// design formulas; I0 is the Abramowitz & Stegun 9.8.1 / 9.8.2 polynomial approximation r8brain uses (r8bbase.h:1216-1236).
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

double sqr( double x ) { return x * x; }

// r8bbase.h:1216-1236 (Abramowitz & Stegun 9.8.1, 9.8.2)
double bessel_i0_as( double x )
	{
	const double ax = std::fabs( x );
	if( ax < 3.75 )
		{
		double y = x / 3.75; y *= y;
		return 1.0 + y * ( 3.5156229 + y * ( 3.0899424 + y * ( 1.2067492 + y * ( 0.2659732 + y * ( 0.360768e-1 + y * 0.45813e-2 ) ) ) ) );
		}
	const double y = 3.75 / ax;
	return std::exp( ax ) / std::sqrt( ax ) * ( 0.39894228 + y * ( 0.1328592e-1 + y * ( 0.225319e-2 + y * ( -0.157565e-2 + y * ( 0.916281e-2 +
		y * ( -0.2057706e-1 + y * ( 0.2635537e-1 + y * ( -0.1647633e-1 + y * 0.392377e-2 ) ) ) ) ) ) ) );
	}

// CDSPFIRFilter::buildLPFilter for ReqTransBand = 2 %, ReqAtten = 206.91 dB, linear phase (the only parameters Flan uses).
// Returns the 2*fl2+1 taps, DC gain `gain`, centre at index fl2.
bool design_default_lowpass( double req_norm_freq, double gain, std::vector<double> & h, int & fl2 )
	{
	const double tb = 2.0 * 0.01;                                              // :229
	double atten = -206.91;                                                    // :233
	atten -= 0.21;                                                             // :268-273 (tb < 0.10, ReqAtten >= 117)
	const int corr_index = int( std::floor( ( -atten - 49.0 ) * 264 / 176.25 + 0.5 ) );   // :285-291
	if( corr_index != 237 ) return false;
	atten -= -19 / 196.0;                                                      // :354-376: entry 237 of the tb < 0.10 correction table is -19, scale 196
	const double pwr = 7.43932822146293e-8 * sqr( atten ) + 0.000102747434588003 * std::cos( 0.00785021930010397 * atten ) *
		std::cos( 0.633854318781239 + 0.103208573657699 * atten ) - 0.00798132247867036 - 0.000903555213543865 * atten -
		0.0969365532127236 * std::exp( 0.0779275237937911 * atten ) - 1.37304948662012e-5 * atten * std::cos( 0.00785021930010397 * atten );   // :379-384
	double hl, fo1;
	if( pwr <= 0.067665322581 )                                                // :386, tb < 0.10 branch :425-435
		{
		hl = ( 2.45739657014937 + 269.183679500541 * pwr * std::cos( 5.73225668178813 +
			std::atan2( std::cosh( 0.988861169868941 - 17.2201556280744 * pwr ), 1.08340138240431 * pwr ) ) ) / tb;
		fo1 = 2.291956939 * tb + 0.01942450693 * sqr( tb ) * hl - 4.67538973161837 * pwr * tb - 1.668433124 * tb * std::pow( pwr, pwr );
		}
	else                                                                       // :450-461
		{
		hl = ( 1.15990238966306 * pwr - 5.02124037125213 * sqr( pwr ) - 0.158676856669827 * atten *
			std::cos( 1.1609073390614 * pwr - 6.33932586197475 * pwr * sqr( pwr ) ) ) / tb;
		fo1 = 0.867344453126885 * tb + 0.052693817907757 * tb * std::log( pwr ) + 0.0895511178735932 * tb * std::atan( 59.7538527741309 * pwr ) -
			0.0745653568081453 * pwr * tb;
		}
	const double pi = 3.14159265358979323846;
	const double len2 = 0.25 * hl / req_norm_freq;                             // :468
	const double freq2 = pi * ( 1.0 - fo1 ) * req_norm_freq;                   // :470
	fl2 = int( std::floor( len2 ) );                                           // CDSPSincFilterGen.h:136
	const double beta = 125.0;                                                 // :464, clamped to [1, 350] (CDSPSincFilterGen.h:584)
	const double kdiv = bessel_i0_as( beta );                                  // CDSPSincFilterGen.h:588
	h.assign( 2 * fl2 + 1, 0.0 );
	auto window = [&]( int t )                                                 // CDSPSincFilterGen.h:246-257, raised to `pwr` (:354, r8bbase.h:1185-1188)
		{
		const double n = 1.0 - sqr( t / len2 );
		const double w = n < 0.0 ? 0.0 : bessel_i0_as( beta * std::sqrt( n ) ) / kdiv;
		return std::pow( w, pwr );
		};
	h[fl2] = freq2 * window( 0 ) / pi;                                         // CDSPSincFilterGen.h:353
	for( int t = 1; t <= fl2; ++t )                                            // :356-365 (Freq1 = 0; the recursive oscillator of the original is sin( freq2 * t ))
		{
		const double v = std::sin( freq2 * t ) * window( t ) / t / pi;
		h[fl2 + t] = v; h[fl2 - t] = v;
		}
	double s = 0.0;                                                            // CDSPFIRFilter.h:496-507: DC gain -> ReqGain
	for( double v : h ) s += v;
	for( double & v : h ) v *= gain / s;
	return true;
	}

} // namespace

extern "C" {

// taps of the default low-pass at cut-off 1/2, for tests
int oracle_r8b_default_lowpass_half( double * taps, int capacity )
	{
	std::vector<double> h; int fl2 = 0;
	if( !design_default_lowpass( 0.5, 1.0, h, fl2 ) ) return -1;
	if( taps ) for( int i = 0; i < int( h.size() ) && i < capacity; ++i ) taps[i] = h[i];
	return int( h.size() );
	}

// AudioConversions.cpp:22  format.num_frames *= new_sample_rate / get_sample_rate()   (int *= float)
int64_t oracle_resample_out_frames( int64_t num_frames, float src_rate, float dst_rate )
	{
	return int64_t( int32_t( float( int32_t( num_frames ) ) * ( dst_rate / src_rate ) ) );
	}

// Audio::resample for src_rate == 2 * dst_rate.  in: float[ch][n] ; out: float[ch][n_out] with n_out = oracle_resample_out_frames.
// The stream is the whole buffer (ch*n samples in, ch*n_out samples out).
int oracle_resample_2to1( const float * in, int64_t total_in, float * out, int64_t total_out )
	{
	std::vector<double> h; int fl2 = 0;
	if( !design_default_lowpass( 0.5, 1.0, h, fl2 ) ) return -1;
	for( int64_t k = 0; k < total_out; ++k )
		{
		double acc = 0.0;
		const int64_t c = 2 * k;
		const int64_t j0 = c - fl2 < 0 ? 0 : c - fl2;
		const int64_t j1 = c + fl2 >= total_in ? total_in - 1 : c + fl2;
		for( int64_t i = j0; i <= j1; ++i ) acc += h[fl2 + ( c - i )] * double( in[i] );
		out[k] = float( acc );
		}
	return 0;
	}

// The single-step rational ratios of CDSPResampler (r8brain/CDSPResampler.h:139-161: src*num == dst*den for (num,den) in
// (1,2) (1,3) (2,3) (3,2) (3,4); :165-207 with c == 0: dst == 2 src, dst == 3 src): ONE CDSPBlockConvolver with the default low-pass
// at cut-off 1/max(up,down) and DC gain `up` (getLPFilter( 1/max, tb, atten, phase, num ), :150-153,:193-195), whole-number
// upsampling by zero stuffing (CDSPBlockConvolver.h copyUpsample), the filter latency consumed, every `down`-th sample kept
// starting with the first (DownSkipInit = 0, InputDelay = 0 for a linear-phase filter):
//        out[k] = float( sum_j h[j] * xu[down*k - j] ),  xu[up*m] = x[m], zero elsewhere and outside the buffer.
// Sum in ascending input index m.  (up,down) = (1,2) is oracle_resample_2to1.
int oracle_resample_rational( const float * in, int64_t total_in, float * out, int64_t total_out, int up, int down )
	{
	if( up < 1 || down < 1 ) return -1;
	std::vector<double> h; int fl2 = 0;
	if( !design_default_lowpass( 1.0 / ( up > down ? up : down ), double( up ), h, fl2 ) ) return -1;
	for( int64_t k = 0; k < total_out; ++k )
		{
		double acc = 0.0;
		const int64_t c = int64_t( down ) * k;                                      // position in the zero-stuffed stream
		// input samples m with |c - up*m| <= fl2
		int64_t m0 = ( c - fl2 + up - 1 ) / up; if( c - fl2 < 0 ) m0 = -( ( fl2 - c ) / up );
		if( m0 < 0 ) m0 = 0;
		int64_t m1 = ( c + fl2 ) / up;
		if( m1 >= total_in ) m1 = total_in - 1;
		for( int64_t m = m0; m <= m1; ++m ) acc += h[fl2 + ( c - up * m )] * double( in[m] );
		out[k] = float( acc );
		}
	return 0;
	}

// taps of the default low-pass at cut-off `req_norm_freq` with DC gain `gain`, for tests; returns the tap count
int oracle_r8b_default_lowpass( double req_norm_freq, double gain, double * taps, int capacity )
	{
	std::vector<double> h; int fl2 = 0;
	if( !design_default_lowpass( req_norm_freq, gain, h, fl2 ) ) return -1;
	if( taps ) for( int i = 0; i < int( h.size() ) && i < capacity; ++i ) taps[i] = h[i];
	return int( h.size() );
	}

} // extern "C"
