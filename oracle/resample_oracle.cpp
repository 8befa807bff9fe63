// resample_oracle.cpp -- CPU restatement of Audio::resample: r8brain's CDSPResampler for the parameters Flan uses.  It grew form by form, and
// every form is kept because the tests hold them against each other: the 2:1 decimation (BASELINE config 5: 96 kHz -> 48 kHz) and the other
// single-convolver ratios; convolver + whole-stepping interpolator (44.1 <-> 48 kHz); chains with half-band stages and the spline-interpolated
// bank (oracle_resample_chain); and the whole constructor as a stage list (oracle_resample_general: what every other form is a special case of).
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
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#include <cstdio>
#include <algorithm>

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

// CDSPFIRFilter::buildLPFilter (CDSPFIRFilter.h:227-493) for ReqAtten = 206.91 dB, linear phase (what Flan always uses) and any transition
// band (2 % everywhere but in the second convolver of the intermediate-interpolation branch, CDSPResampler.h:296-302).
// Returns the 2*fl2+1 taps, DC gain `gain`, centre at index fl2.
bool design_lowpass( double req_norm_freq, double tb_percent, double gain, std::vector<double> & h, int & fl2 )
	{
	const double tb = tb_percent * 0.01;                                       // :229
	if( !( tb > 0.0 ) ) return false;
	double atten = -206.91;                                                    // :233
	atten -= tb >= 0.25 ? 1.60 : tb >= 0.10 ? 0.69 : 0.21;                     // :235-281 (the ReqAtten >= 117 rows)
	const int corr_index = int( std::floor( ( -atten - 49.0 ) * 264 / 176.25 + 0.5 ) );   // :283-291
	// :293-376: the one entry of each correction table this attenuation lands on (index 239 / 238 / 237)
	if( corr_index != ( tb >= 0.25 ? 239 : tb >= 0.10 ? 238 : 237 ) ) return false;
	atten -= tb >= 0.25 ? -12 / 101.0 : tb >= 0.10 ? -62 / 210.0 : -19 / 196.0;
	const double pwr = 7.43932822146293e-8 * sqr( atten ) + 0.000102747434588003 * std::cos( 0.00785021930010397 * atten ) *
		std::cos( 0.633854318781239 + 0.103208573657699 * atten ) - 0.00798132247867036 - 0.000903555213543865 * atten -
		0.0969365532127236 * std::exp( 0.0779275237937911 * atten ) - 1.37304948662012e-5 * atten * std::cos( 0.00785021930010397 * atten );   // :379-384
	double hl, fo1;
	if( pwr <= 0.067665322581 )                                                // :386
		{
		if( tb >= 0.25 )                                                       // :388-401
			{
			hl = 2.6778150875894 / tb + 300.547590563091 * std::atan( std::atan( 2.68959772209918 * pwr ) ) /
				( 5.5099277187035 * tb - tb * std::tanh( std::cos( std::asinh( atten ) ) ) );
			fo1 = 0.987205355829873 * tb + 1.00011788929851 * std::atan2( -0.321432067051302 - 6.19131357321578 * std::sqrt( pwr ),
				hl + -1.14861472207245 / ( hl - 14.1821147585957 ) + std::pow( 0.9521145021664, std::pow( std::atan2( 1.12018764830637, tb ),
				2.10988901686912 * hl - 20.9691278378345 ) ) );
			}
		else if( tb >= 0.10 )                                                  // :403-414
			{
			hl = ( 1.56688617018066 + 142.064321294568 * pwr + 0.00419441117131136 * std::cos( 243.633511747297 * pwr ) -
				0.022953443903576 * atten - 0.026629568860284 * std::cos( 127.715550622571 * pwr ) ) / tb;
			fo1 = 0.982299356642411 * tb + 0.999441744774215 * std::asinh( ( -0.361783054039583 - 5.80540593623676 * std::sqrt( pwr ) ) / hl );
			}
		else                                                                   // :416-426
			{
			hl = ( 2.45739657014937 + 269.183679500541 * pwr * std::cos( 5.73225668178813 +
				std::atan2( std::cosh( 0.988861169868941 - 17.2201556280744 * pwr ), 1.08340138240431 * pwr ) ) ) / tb;
			fo1 = 2.291956939 * tb + 0.01942450693 * sqr( tb ) * hl - 4.67538973161837 * pwr * tb - 1.668433124 * tb * std::pow( pwr, pwr );
			}
		}
	else
		{
		if( tb >= 0.25 )                                                       // :430-438
			{
			hl = ( 1.50258368698213 + 158.556968859477 * std::asinh( pwr ) * std::tanh( 57.9466246871383 * std::tanh( pwr ) ) - 0.0105440479814834 * atten ) / tb;
			fo1 = 0.994024401639321 * tb + ( -0.236282717577215 - 6.8724924545387 * std::sqrt( std::sin( pwr ) ) ) / hl;
			}
		else if( tb >= 0.10 )                                                  // :440-449
			{
			hl = ( 1.50277377248945 + 158.222625721046 * std::asinh( pwr ) * std::tanh( 1.02875299001715 + 42.072277322604 * pwr ) - 0.0108380943845632 * atten ) / tb;
			fo1 = 0.992539376734551 * tb + ( -0.251747813037178 - 6.74159892452584 * std::sqrt( std::tanh( std::tanh( std::tan( pwr ) ) ) ) ) / hl;
			}
		else                                                                   // :450-461
			{
			hl = ( 1.15990238966306 * pwr - 5.02124037125213 * sqr( pwr ) - 0.158676856669827 * atten *
				std::cos( 1.1609073390614 * pwr - 6.33932586197475 * pwr * sqr( pwr ) ) ) / tb;
			fo1 = 0.867344453126885 * tb + 0.052693817907757 * tb * std::log( pwr ) + 0.0895511178735932 * tb * std::atan( 59.7538527741309 * pwr ) -
				0.0745653568081453 * pwr * tb;
			}
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

bool design_default_lowpass( double req_norm_freq, double gain, std::vector<double> & h, int & fl2 ) { return design_lowpass( req_norm_freq, 2.0, gain, h, fl2 ); }

} // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// The two-stage ratios (44.1 <-> 48 kHz and every other pair CDSPResampler serves with ONE block convolver followed by ONE
// CDSPFracInterpolator, no half-band stages):
//   * dst*2 > src, not a single-step ratio (CDSPResampler.h:214-316 with c == 0, which includes c == 1 turned into 0 by whole stepping,
//     :268-276): CDSPBlockConvolver( getLPFilter( NormFreq, tb, atten, phase, 2.0 ), 2, 1 ) with NormFreq = 0.5 when upsampling and
//     0.5*dst/src otherwise, then CDSPFracInterpolator( 2 src -> dst, IsThird = false );
//   * 2 dst <= src < 4 dst, src not 2 dst or 3 dst (:319-378 with c == 0, UseInterp): CDSPBlockConvolver( getLPFilter( dst/src, ...,
//     1.0 ), 1, 1 ), then CDSPFracInterpolator( src -> dst, IsThird = ( 3 dst <= src ) ).
// Both filters are linear phase, so no fractional latency travels between the stages (CDSPBlockConvolver.h LatencyFrac = 0) and the
// interpolator starts at position 0.  Only WHOLE-NUMBER STEPPING (getWholeStepping, CDSPFracInterpolator.h:573-602) is restated: the
// interpolator then walks a bank of OutStep fractional-delay filters with exact integer positions (convolve0, :929-958); the
// spline-interpolated bank of the other case (convolve2) re-bases its position counter at process() call boundaries and is not restated.
namespace {

// CDSPFracInterpolator.h:539-558
bool find_gcd( double l, double s, double & gcd )
	{
	for( int it = 0; it < 50; ++it )
		{
		if( s <= 0.0 ) { gcd = l; return true; }
		const double r = l - s;
		l = s;
		s = r < 0.0 ? -r : r;
		}
	return false;
	}

// CDSPFracInterpolator.h:573-602
bool whole_stepping( double src, double dst, int & in_step, int & out_step )
	{
	double gcd;
	if( !find_gcd( src, dst, gcd ) || gcd < 1.0 ) return false;
	const double i0 = src / gcd, o0 = dst / gcd;
	in_step = int( i0 ); out_step = int( o0 );
	if( i0 != in_step || o0 != out_step ) return false;
	return out_step <= 1500;
	}

struct TwoStage { int up; double norm_freq, gain; bool third; int in_step, out_step; };

// which chain does CDSPResampler( src, dst ) build?  false: not one of the two-stage shapes above
bool two_stage_shape( double src, double dst, TwoStage & ts )
	{
	if( src == dst ) return false;
	static const int common[5][2] = { { 1, 2 }, { 1, 3 }, { 2, 3 }, { 3, 2 }, { 3, 4 } };            // CDSPResampler.h:142-170
	for( const auto & c : common ) if( src * c[0] == dst * c[1] ) return false;
	for( int i = 2; i <= 3; ++i )                                                                   // :174-212
		for( int c = 0; src * ( i << c ) <= dst; ++c ) if( src * ( i << c ) == dst ) return false;
	if( dst * 2 > src )                                                                            // :214
		{
		const double thresh = src * 1.01;                                                          // :229-244
		int c = 0, div = 1;
		while( !( dst < thresh * ( div * 2 ) ) ) { div *= 2; ++c; }
		int t1, t2;
		if( c == 1 && whole_stepping( src * 2.0, dst, t1, t2 ) ) c = 0;                            // :266-276
		if( c > 0 ) return false;                                                                  // half-band upsamplers follow
		ts.up = 2; ts.norm_freq = dst > src ? 0.5 : 0.5 * dst / src; ts.gain = 2.0; ts.third = false;   // :218-225, :312-313
		return whole_stepping( src * 2.0, dst, ts.in_step, ts.out_step );
		}
	if( dst * 4.0 <= src ) return false;                                                           // :321-331: c > 0, half-band downsamplers
	for( int downf = 2; downf <= 3; ++downf ) if( dst * downf == src ) return false;               // :340-349 (single step; matched above anyway)
	ts.up = 1; ts.norm_freq = dst / src; ts.gain = 1.0; ts.third = ts.norm_freq * 3.0 <= 1.0;      // :351-356, :368-376
	return whole_stepping( src, dst, ts.in_step, ts.out_step );
	}

// CDSPFracDelayFilterBank::getWinParams (CDSPFracInterpolator.h:289-348): Kaiser beta / window power by attenuation, filter length
const double * frac_win_params( double att, bool third, int & flt_len )
	{
	static const double half[13][3] = {
		{ 2.6504246356892924, 1.9035845248358245, 51.7280 }, { 4.0759654812373016, 1.5747323142948524, 67.1095 },
		{ 4.9036508646352033, 1.6207644759455790, 81.8379 }, { 5.6131421124830716, 1.6947677220415129, 96.4021 },
		{ 5.9433751253133691, 1.8730186383321272, 111.1300 }, { 6.8308658253825660, 1.8549555120377224, 125.4649 },
		{ 7.6648458853758372, 1.8565765953924642, 139.7378 }, { 8.2038730802326842, 1.9269521308895179, 154.0532 },
		{ 8.7865151489187561, 1.9775307528231671, 168.2101 }, { 9.5945013206755156, 1.9718457932433306, 182.1076 },
		{ 10.5163048616210250, 1.9504085061576968, 195.5668 }, { 10.2382664677006100, 2.1608878780497056, 209.0609 },
		{ 10.9976663155261660, 2.1536415815428249, 222.5009 } };
	static const double thirds[10][3] = {
		{ 4.0738201365282452, 1.5774150265957998, 67.2431 }, { 4.9502289040040495, 1.7149006172407628, 86.4870 },
		{ 5.5995071332976192, 1.8930163359641823, 106.1171 }, { 6.3627287856776054, 1.9945748303811506, 125.2304 },
		{ 7.4299554386534528, 1.9893399585993299, 144.3469 }, { 8.0667710807396436, 2.0928202837610885, 163.4098 },
		{ 8.7469991933128526, 2.1640274270903488, 181.0694 }, { 10.0823164330540570, 2.0896732996403280, 199.2880 },
		{ 19.1718281840114810, 1.2030083075440616, 215.2990 }, { 21.0914128488567630, 1.1919045429676862, 233.9152 } };
	int i = 0;
	if( third ) { while( i != 9 && thirds[i][2] < att ) ++i; flt_len = ( i + 3 ) * 2; return thirds[i]; }
	while( i != 12 && half[i][2] < att ) ++i;
	flt_len = ( i + 3 ) * 2;
	return half[i];
	}

// One fractional-delay filter of the bank: CDSPSincFilterGen::initFrac + generateFrac with the Kaiser-power window
// (CDSPSincFilterGen.h:184-193, :246-257, :432-517, :572-590) and normalizeFIRFilter( ., 1.0 ) (r8bbase.h:943-970).
// Len2 = flt_len / 2 is a whole number here, so fl2 = Len2 and the first tap never falls outside the window (:444-450).
void frac_delay_filter( int flt_len, double beta, double power, double delay, double * op )
	{
	const double pi = 3.14159265358979323846;
	const int fl2 = flt_len / 2;
	const double len2 = fl2;
	const double kdiv = bessel_i0_as( beta ), len2frac = delay / len2;
	int wn = -fl2;
	auto win = [&]()
		{
		const double n = 1.0 - sqr( wn / len2 + len2frac );
		++wn;
		const double w = n < 0.0 ? 0.0 : bessel_i0_as( beta * std::sqrt( n ) ) / kdiv;
		return std::pow( w, power );
		};
	const double f[2] = { std::sin( delay * pi ), -std::sin( delay * pi ) };
	int t = -fl2;
	if( t + delay < -len2 ) { (void) win(); *op++ = 0.0; ++t; }                 // :444-450: a negative delay (the spline bank's rows beyond 1) pushes the first tap out of the window
	const int mt = ( delay >= 1.0 - 1e-13 && delay <= 1.0 + 1e-13 ) ? -1 : 0;
	for( ; t < mt; ++t ) *op++ = f[t & 1] * win() / ( t + delay ) / pi;
	double ut = t + delay;
	*op = std::fabs( ut ) <= 1e-13 ? win() : f[t & 1] * win() / ut / pi;
	while( t < fl2 - 2 ) { ++op; ++t; *op = f[t & 1] * win() / ( t + delay ) / pi; }
	++op; ++t;
	ut = t + delay;
	*op = ut > len2 ? 0.0 : f[t & 1] * win() / ut / pi;
	op -= flt_len - 1;
	double s = 0.0;
	for( int i = 0; i < flt_len; ++i ) s += op[i];
	s = 1.0 / s;
	for( int i = 0; i < flt_len; ++i ) op[i] *= s;
	}

// CDSPFracDelayFilterBank( OutStep, 1, 2, 206.91, third ) (CDSPFracInterpolator.h:64-121): row r is the delay ( fracs - r ) / fracs
void frac_delay_bank( int fracs, bool third, std::vector<double> & bank, int & flt_len )
	{
	const double * wp = frac_win_params( 206.91, third, flt_len );
	const double beta = wp[0] < 1.0 ? 1.0 : wp[0] > 350.0 ? 350.0 : wp[0];
	bank.assign( size_t( fracs ) * flt_len, 0.0 );
	for( int r = 0; r < fracs; ++r ) frac_delay_filter( flt_len, beta, std::fabs( wp[1] ), double( fracs - r ) / fracs, &bank[size_t( r ) * flt_len] );
	}

// ---- the interpolator WITHOUT whole stepping (CDSPFracInterpolator.h: getFilterBank( -1, 3, 8, ... ), convolve2 :960-1005) ---------------
// The bank: FilterFracs = ceil( 1.792462178761753 exp( 0.033300466782047 ReqAtten ) ) rows plus 8 for the spline's reach (:97-100, :115-127:
// row i has the delay ( FilterFracs - i ) / FilterFracs for i = -3 .. FilterFracs + 4), then every tap of rows 0 .. FilterFracs replaced in
// place by the 8-point 2nd-order spline through that tap of 8 consecutive rows (r8bbase.h:1019-1029): { x0, c1, c2 } per tap.
// (the cache rounds the requested attenuation UP to its table row first -- roundReqAtten, :394-398, :151-155 -- 209.0609 dB for the half-band
// parameters, 215.2990 dB for the third-band ones: 1893 / 2329 rows)
int spline_bank_fracs( bool third )
	{
	int flt_len; const double * wp = frac_win_params( 206.91, third, flt_len );
	return int( std::ceil( 1.792462178761753 * std::exp( 0.033300466782047 * wp[2] ) ) );
	}

void frac_spline_bank( bool third, std::vector<double> & bank, int & flt_len, int & fracs )
	{
	fracs = spline_bank_fracs( third );
	const double * wp = frac_win_params( 206.91, third, flt_len );
	const double beta = wp[0] < 1.0 ? 1.0 : wp[0] > 350.0 ? 350.0 : wp[0];
	const int rows = fracs + 8;
	std::vector<double> raw( size_t( rows ) * flt_len );
	for( int r = 0; r < rows; ++r ) frac_delay_filter( flt_len, beta, std::fabs( wp[1] ), double( fracs - ( r - 3 ) ) / fracs, &raw[size_t( r ) * flt_len] );
	bank.assign( size_t( fracs + 1 ) * flt_len * 3, 0.0 );
	for( int r = 0; r <= fracs; ++r )
		for( int i = 0; i < flt_len; ++i )
			{
			auto x = [&]( int d ) { return raw[size_t( r + d ) * flt_len + i]; };   // xm3 = row r ... x4 = row r + 7
			double * c = &bank[( size_t( r ) * flt_len + i ) * 3];
			c[0] = x( 3 );
			c[1] = ( 61.0 * ( x( 4 ) - x( 2 ) ) + 16.0 * ( x( 1 ) - x( 5 ) ) + 3.0 * ( x( 6 ) - x( 0 ) ) ) / 76.0;
			c[2] = ( 106.0 * ( x( 2 ) + x( 4 ) ) + 10.0 * x( 6 ) + 6.0 * x( 0 ) - 3.0 * x( 7 ) - 29.0 * ( x( 1 ) + x( 5 ) ) - 167.0 * x( 3 ) ) / 76.0;
			}
	}

// CDSPBlockConvolver's consumed latency (CDSPBlockConvolver.h:62-100 with a power-of-two UpFactor and DownFactor 1; CDSPFIRFilter.h:467-478):
// the stream y it hands on after n_in input samples has max( 0, up n_in - latency ) samples
int64_t block_convolver_latency( int fl2, int up )
	{
	const int kernel_len = 2 * fl2 + 1;
	int bits = 0; while( ( ( kernel_len - 1 ) >> bits ) != 0 ) ++bits;            // getBitOccupancy (r8bbase.h:775-)
	if( bits < 1 ) bits = 1;
	const int block_len2 = 2 << bits;
	const int prev_input = ( kernel_len - 1 + up - 1 ) / up;
	return int64_t( block_len2 - prev_input * up ) + fl2;                          // InputLen + the filter's latency
	}

} // namespace

// 1 when ( src, dst ) is a two-stage ratio this file restates; fills what a test wants to look at
extern "C" int oracle_resample_two_stage_shape( double src, double dst, int * up, double * norm_freq, int * third, int * in_step, int * out_step )
	{
	TwoStage ts{};
	if( !two_stage_shape( src, dst, ts ) ) return 0;
	if( up ) *up = ts.up;
	if( norm_freq ) *norm_freq = ts.norm_freq;
	if( third ) *third = ts.third ? 1 : 0;
	if( in_step ) *in_step = ts.in_step;
	if( out_step ) *out_step = ts.out_step;
	return 1;
	}

// the bank, for tests: returns the filter length, rows [out_step][flt_len]
extern "C" int oracle_r8b_frac_bank( int fracs, int third, double * rows, int capacity )
	{
	std::vector<double> bank; int flt_len = 0;
	frac_delay_bank( fracs, third != 0, bank, flt_len );
	if( rows ) for( size_t i = 0; i < bank.size() && i < size_t( capacity ); ++i ) rows[i] = bank[i];
	return flt_len;
	}

// Audio::resample for a two-stage ratio.  Stage 1 (block convolver, latency consumed; the direct sum as above):
//        y[n] = sum_m h[fl2 + n - up*m] * x[m],  m ascending, x = 0 outside the buffer, y = 0 for n < 0 (the interpolator's ring buffer
//        starts with fl2i - 1 zeros, CDSPFracInterpolator.h:771-778);
// stage 2 (convolve0): output k sits at k*in_step/out_step input samples: p = ( k*in_step ) / out_step, r = ( k*in_step ) % out_step,
//        out[k] = float( sum_{i=0}^{flt_len-1} bank[r][i] * y[p - ( fl2i - 1 ) + i] ),  i ascending, separate multiply and add.
extern "C" int oracle_resample_two_stage( const float * in, int64_t total_in, float * out, int64_t total_out, double src, double dst )
	{
	TwoStage ts{};
	if( !two_stage_shape( src, dst, ts ) ) return -1;
	std::vector<double> h; int fl2 = 0;
	if( !design_default_lowpass( ts.norm_freq, ts.gain, h, fl2 ) ) return -1;
	std::vector<double> bank; int flt_len = 0;
	frac_delay_bank( ts.out_step, ts.third, bank, flt_len );
	const int fll = flt_len / 2 - 1;
	if( total_out <= 0 ) return 0;
	const int64_t ny = ( ( total_out - 1 ) * ts.in_step ) / ts.out_step - fll + flt_len;
	std::vector<double> y( size_t( ny > 0 ? ny : 0 ) );
	const int up = ts.up;
	for( int64_t n = 0; n < ny; ++n )
		{
		int64_t m0 = n - fl2 <= 0 ? 0 : ( n - fl2 + up - 1 ) / up;
		int64_t m1 = ( n + fl2 ) / up;
		if( m1 >= total_in ) m1 = total_in - 1;
		double acc = 0.0;
		for( int64_t m = m0; m <= m1; ++m ) acc += h[fl2 + ( n - up * m )] * double( in[m] );
		y[size_t( n )] = acc;
		}
	for( int64_t k = 0; k < total_out; ++k )
		{
		const int64_t pos = k * ts.in_step, p = pos / ts.out_step;
		const double * ft = &bank[size_t( pos % ts.out_step ) * flt_len];
		double s = 0.0;
		for( int i = 0; i < flt_len; ++i )
			{
			const int64_t a = p - fll + i;
			s += ft[i] * ( a >= 0 && a < ny ? y[size_t( a )] : 0.0 );
			}
		out[k] = float( s );
		}
	return 0;
	}

// ---------------------------------------------------------------------------------------------------------------------------------
// Chains with half-band stages (r8brain/CDSPResampler.h:174-212: dst = 2^c * i * src, i = 2 or 3, c >= 1; :319-378 with c >= 1:
// src >= 4 dst).  CDSPHBUpsampler (CDSPHBUpsampler.h:560-720): out[2j] = in[j], out[2j+1] = sum_t flt[t] ( in[j+1+t] + in[j-t] );
// CDSPHBDownsampler (CDSPHBDownsampler.h:95-260): out[j] = in[2j] + sum_t flt[t] ( in[2j+2t+1] + in[2j-2t-1] ); both zero latency,
// in = 0 before the stream starts.  The kernels are the ones getHBFilter / getHBFilterThird pick at ReqAtten = 206.91 dB
// (CDSPHBUpsampler.h:43-215, :296-436), by SteepIndex.
namespace {

const double * hb_kernel( int steep, bool third, int & n )
	{
	static const double h0[13] = { 6.2816416238782957e-001, -1.8809076918442266e-001, 9.0918539368474965e-002, -4.6765502172995604e-002,      // HBKernel_13, 215.1364 dB
		2.3287520069933797e-002, -1.0760626940880943e-002, 4.4853921118213676e-003, -1.6438774496992904e-003, 5.1441308429384374e-004,
		-1.3211724349740752e-004, 2.6191316362108199e-005, -3.5802424384280469e-006, 2.5491272423372411e-007 };
	static const double h1[7] = { 6.1610372237019151e-001, -1.5767891821295410e-001, 5.5089690570484962e-002, -1.6895755290596615e-002,       // HBKernel_7b, 209.9472 dB
		3.9416641999499014e-003, -6.0603620400878633e-004, 4.5632598748568398e-005 };
	static const double h2[5] = { 6.0626808278478261e-001, -1.3588224019070938e-001, 3.5544305138258458e-002, -6.5127022013993230e-003,       // HBKernel_5c, 213.4984 dB
		5.8255449020627736e-004 };
	static const double h3[4] = { 5.9835028661892165e-001, -1.1999986095168852e-001, 2.4132530901858028e-002, -2.4829565783680927e-003 };     // HBKernel_4d, 220.6519 dB
	static const double t0[9] = { 6.2163188987470752e-001, -1.7108115412330563e-001, 6.9588371105224839e-002, -2.7339625869282957e-002,       // third: HBKernel_9, 220.5199 dB
		9.2954473703765472e-003, -2.5537181861669997e-003, 5.2572296540671394e-004, -7.1813366796731157e-005, 4.8802392556669750e-006 };
	static const double t1[6] = { 6.1161456377889145e-001, -1.4743902036519768e-001, 4.5344160828746795e-002, -1.1207372108402218e-002,       // third: HBKernel_6b, 224.2705 dB
		1.8328498006058664e-003, -1.4518194076022933e-004 };
	static const double t2[5] = { 6.0590922849004858e-001, -1.3515953371903033e-001, 3.5020856634677522e-002, -6.3256195330255094e-003,       // third: HBKernel_5c, 248.8728 dB
		5.5506812768978109e-004 };
	static const double h45[4] = { 5.9819599535791312e-001, -1.1972157884617740e-001, 2.3977307400990484e-002, -2.4517239127622593e-003 };    // HBKernel_4e, 268.8561 dB (SteepIndex 4, 5)
	static const double h6[3] = { 5.8594191093025305e-001, -9.7662866644414148e-002, 1.1720955714177778e-002 };                               // HBKernel_3g, 275.5531 dB (6 and beyond)
	static const double t3[4] = { 5.9823601283411165e-001, -1.1979369067338455e-001, 2.4017459011435899e-002, -2.4597811725236445e-003 };     // third: HBKernel_4d, 248.8578 dB
	static const double t45[3] = { 5.8596887233874539e-001, -9.7703321108182931e-002, 1.1734448775437802e-002 };                              // third: HBKernel_3e, 224.4366 dB
	static const double t6[3] = { 5.8593945769687561e-001, -9.7659186594368730e-002, 1.1719728897494584e-002 };                               // third: HBKernel_3g, 296.4833 dB
	if( steep < 0 ) { n = 0; return nullptr; }
	if( third ) switch( steep ) { case 0: n = 9; return t0; case 1: n = 6; return t1; case 2: n = 5; return t2; case 3: n = 4; return t3; case 4: case 5: n = 3; return t45; default: n = 3; return t6; }
	switch( steep ) { case 0: n = 13; return h0; case 1: n = 7; return h1; case 2: n = 5; return h2; case 3: n = 4; return h3; case 4: case 5: n = 4; return h45; default: n = 3; return h6; }
	}

// what CDSPResampler( src, dst ) builds, for the shapes this file restates
struct Chain
	{
	int hb_down = 0;                       // CDSPHBDownsampler stages in front (SteepIndex hb_down-1 .. 0)
	int up = 1, down = 1; double norm_freq = 0.5, gain = 1.0;    // the block convolver
	int hb_up = 0;                         // CDSPHBUpsampler stages behind it (SteepIndex 0 .. hb_up-1)
	bool third = false;                    // third-band half-band kernels / interpolator bank
	bool interp = false; int in_step = 0, out_step = 0;          // whole-stepping CDSPFracInterpolator last ...
	bool spline = false; double isrc = 0.0, idst = 0.0;          // ... or the spline-interpolated bank when the rates have no whole stepping
	};

bool chain_shape( double src, double dst, Chain & ch )
	{
	if( src == dst ) return false;
	static const int common[5][2] = { { 1, 2 }, { 1, 3 }, { 2, 3 }, { 3, 2 }, { 3, 4 } };            // CDSPResampler.h:142-170
	for( const auto & c : common )
		if( src * c[0] == dst * c[1] ) { ch.up = c[0]; ch.down = c[1]; ch.norm_freq = 1.0 / ( c[0] > c[1] ? c[0] : c[1] ); ch.gain = c[0]; return true; }
	for( int i = 2; i <= 3; ++i )                                                                   // :174-212
		for( int c = 0; src * ( i << c ) <= dst; ++c )
			if( src * ( i << c ) == dst )
				{
				ch.up = i; ch.down = 1; ch.norm_freq = 1.0 / i; ch.gain = i; ch.hb_up = c; ch.third = ( i == 3 );
				int n; return c == 0 || hb_kernel( c - 1, ch.third, n ) != nullptr;
				}
	TwoStage ts{};
	if( dst * 2 > src )                                                                            // :214-316
		{
		if( two_stage_shape( src, dst, ts ) )
			{
			ch.up = ts.up; ch.norm_freq = ts.norm_freq; ch.gain = ts.gain; ch.third = ts.third; ch.interp = true; ch.in_step = ts.in_step; ch.out_step = ts.out_step;
			return true;
			}
		// no whole stepping from 2 src: the same chain with the spline-interpolated bank, unless intermediate interpolation is chosen (:229-276)
		const double thresh = src * 1.01;
		if( !( dst < thresh * 2 ) ) return false;
		ch.up = 2; ch.norm_freq = dst > src ? 0.5 : 0.5 * dst / src; ch.gain = 2.0; ch.interp = true; ch.spline = true; ch.isrc = src * 2.0; ch.idst = dst;
		return true;
		}
	double check = dst * 4.0, fin_gain = 1.0;                                                      // :319-331
	int c = 0;
	while( check <= src ) { ++c; check *= 2.0; fin_gain *= 0.5; }
	const int div = 1 << c;
	ch.hb_down = c; ch.gain = fin_gain; ch.up = 1;
	int downf = 0;
	for( int d = 2; d <= 3; ++d ) if( dst * div * d == src ) { downf = d; break; }                 // :340-349
	if( downf ) { ch.down = downf; ch.norm_freq = 1.0 / downf; ch.third = ( downf == 3 ); }
	else                                                                                           // :351-356, :372-376
		{
		ch.down = 1; ch.norm_freq = dst * div / src; ch.third = ch.norm_freq * 3.0 <= 1.0;
		ch.interp = true;
		if( !whole_stepping( src, dst * div, ch.in_step, ch.out_step ) )
			{
			if( c > 0 ) return false;                                                              // (half-band stages in front of the spline bank: not restated)
			ch.spline = true; ch.isrc = src; ch.idst = dst * div;
			}
		}
	int n; return c == 0 || hb_kernel( c - 1, ch.third, n ) != nullptr;
	}

double at( const std::vector<double> & v, int64_t i ) { return i >= 0 && i < int64_t( v.size() ) ? v[size_t( i )] : 0.0; }

} // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// The whole of CDSPResampler's constructor (CDSPResampler.h:119-378) as a list of stages, and a stage-by-stage evaluation on the whole
// stream.  Adds to the chains above: the upsampling branch with INTERMEDIATE INTERPOLATION (dst >= 2.02 src off the 2^k / 3 2^k grid, :229-310:
// 2x convolver -> interpolator to dst / 2^c -> a 2x convolver whose transition band follows from the rates -> c - 1 half-band upsamplers) and
// half-band downsamplers in front of the spline-interpolated bank.
namespace {

struct Stage
	{
	enum Kind { HbDown, Conv, HbUp, Frac } kind;
	int steep = 0; bool third = false;                                            // half-band stages; the interpolator's bank
	int up = 1, down = 1; double nf = 0.5, tb = 2.0, gain = 1.0;                  // block convolver (tb in percent)
	bool whole = true; int in_step = 0, out_step = 0; double isrc = 0.0, idst = 0.0;   // interpolator
	};

bool add_frac( std::vector<Stage> & st, double isrc, double idst, bool third )
	{
	Stage f; f.kind = Stage::Frac; f.third = third; f.isrc = isrc; f.idst = idst;
	f.whole = whole_stepping( isrc, idst, f.in_step, f.out_step );
	st.push_back( f );
	return true;
	}

bool build_stages( double src, double dst, std::vector<Stage> & st )
	{
	st.clear();
	if( !( src > 0.0 ) || !( dst > 0.0 ) || src == dst ) return false;
	auto conv = [&]( int up, int down, double nf, double tb, double gain ) { Stage c; c.kind = Stage::Conv; c.up = up; c.down = down; c.nf = nf; c.tb = tb; c.gain = gain; st.push_back( c ); };
	auto hb = [&]( Stage::Kind k, int steep, bool third ) { Stage h; h.kind = k; h.steep = steep; h.third = third; st.push_back( h ); int n; return hb_kernel( steep, third, n ) != nullptr; };
	static const int common[5][2] = { { 1, 2 }, { 1, 3 }, { 2, 3 }, { 3, 2 }, { 3, 4 } };            // :142-170
	for( const auto & c : common )
		if( src * c[0] == dst * c[1] ) { conv( c[0], c[1], 1.0 / ( c[0] > c[1] ? c[0] : c[1] ), 2.0, c[0] ); return true; }
	for( int i = 2; i <= 3; ++i )                                                                   // :174-212
		for( int c = 0; src * ( i << c ) <= dst; ++c )
			if( src * ( i << c ) == dst )
				{
				conv( i, 1, 1.0 / i, 2.0, i );
				for( int k = 0; k < c; ++k ) if( !hb( Stage::HbUp, k, i == 3 ) ) return false;
				return true;
				}
	if( dst * 2 > src )                                                                            // :214-316
		{
		conv( 2, 1, dst > src ? 0.5 : 0.5 * dst / src, 2.0, 2.0 );                                 // :218-225
		const double thresh = src * 1.01;
		int c = 0, div = 1;
		while( !( dst < thresh * ( div * 2 ) ) ) { div *= 2; ++c; }                                // :229-244
		// (:246-261: the 3x variant tries 3 div against the same threshold the loop above just failed at 2 div -- it never wins)
		int t1, t2;
		if( c == 1 && whole_stepping( src * 2.0, dst, t1, t2 ) ) c = 0;                            // :266-276
		if( c > 0 )
			{
			add_frac( st, src * 2.0 * div, dst, false );                                           // :293-294
			conv( 2, 1, 0.5, 100.0 * ( 1.0 - src * div / dst ) / 1.75, 2.0 );                      // :296-302
			for( int i = 1; i < c; ++i ) if( !hb( Stage::HbUp, i - 1, false ) ) return false;      // :304-308
			}
		else add_frac( st, src * 2.0, dst, false );                                                // :312-313
		return true;
		}
	double check = dst * 4.0, fin_gain = 1.0;                                                      // :319-331
	int c = 0;
	while( check <= src ) { ++c; check *= 2.0; fin_gain *= 0.5; }
	const int div = 1 << c;
	int downf = 1; double nf = 0.5; bool use_interp = true, third = false;
	for( int d = 2; d <= 3; ++d ) if( dst * div * d == src ) { downf = d; nf = 1.0 / d; use_interp = false; third = ( d == 3 ); break; }   // :340-349
	if( use_interp ) { nf = dst * div / src; third = nf * 3.0 <= 1.0; }                            // :351-356
	for( int i = 0; i < c; ++i ) if( !hb( Stage::HbDown, c - 1 - i, third ) ) return false;        // :358-365
	conv( 1, downf, nf, 2.0, fin_gain );                                                           // :367-370
	if( use_interp ) add_frac( st, src, dst * div, third );                                        // :372-376
	return true;
	}

// samples a stage has delivered once `n` samples have gone into it (every stage consumes its latency and then keeps pace with its input):
// what decides how much of the stream each process() call hands the interpolator
int64_t delivered( const Stage & s, int64_t n, int fl2_conv )
	{
	if( s.kind == Stage::HbDown )                                                                  // CDSPHBDownsampler.h:95-150: fl2 = 2 taps - 1
		{
		int taps; hb_kernel( s.steep, s.third, taps );
		const int fl2 = 2 * taps - 1;
		return n > fl2 ? ( n - fl2 + 1 ) >> 1 : 0;
		}
	if( s.kind == Stage::Conv ) return std::max<int64_t>( 0, n * s.up - block_convolver_latency( fl2_conv, s.up ) ) / s.down;
	return n;
	}

int resample_by_stages( const std::vector<Stage> & st, const float * in, int64_t total_in, float * out, int64_t total_out, int64_t chunk )
	{
	if( total_out <= 0 ) return 0;
	const int ns = int( st.size() );
	// filters and banks
	std::vector<std::vector<double>> taps; taps.resize( size_t( ns ) ); std::vector<int> fl2( size_t( ns ), 0 ), flt_len( size_t( ns ), 0 ), fracs( size_t( ns ), 0 );
	for( int k = 0; k < ns; ++k )
		{
		if( st[k].kind == Stage::Conv && !design_lowpass( st[k].nf, st[k].tb, st[k].gain, taps[k], fl2[k] ) ) return -1;
		if( st[k].kind == Stage::Frac ) { if( st[k].whole ) { frac_delay_bank( st[k].out_step, st[k].third, taps[k], flt_len[k] ); fracs[k] = st[k].out_step; } else frac_spline_bank( st[k].third, taps[k], flt_len[k], fracs[k] ); }
		}
	// samples each stage must deliver (backwards)
	std::vector<int64_t> need; need.resize( size_t( ns ) );
	need[ns - 1] = total_out;
	for( int k = ns - 1; k > 0; --k )
		{
		const Stage & s = st[k];
		const int64_t n = need[k];
		int64_t r = n;
		if( s.kind == Stage::HbDown ) { int t; hb_kernel( s.steep, s.third, t ); r = 2 * ( n - 1 ) + 2 * t; }
		else if( s.kind == Stage::HbUp ) { int t; hb_kernel( s.steep, s.third, t ); r = ( n - 1 ) / 2 + t + 1; }
		else if( s.kind == Stage::Conv ) r = ( int64_t( s.down ) * ( n - 1 ) + fl2[k] ) / s.up + 1;
		else if( s.whole ) r = ( ( n - 1 ) * s.in_step ) / s.out_step - ( flt_len[k] / 2 - 1 ) + flt_len[k];
		else r = int64_t( std::ceil( double( n ) * s.isrc / s.idst ) ) + flt_len[k] + 8;
		need[k - 1] = std::max<int64_t>( r, 1 );
		}
	std::vector<double> cur( in, in + total_in );
	for( int k = 0; k < ns; ++k )
		{
		const Stage & s = st[k];
		std::vector<double> nxt( size_t( need[k] ) );
		const int64_t n_out = need[k];
		if( s.kind == Stage::HbDown )
			{
			int t; const double * flt = hb_kernel( s.steep, s.third, t );
			for( int64_t j = 0; j < n_out; ++j )
				{
				double a = at( cur, 2 * j );
				for( int i = 0; i < t; ++i ) a += flt[i] * ( at( cur, 2 * j + 2 * i + 1 ) + at( cur, 2 * j - 2 * i - 1 ) );
				nxt[size_t( j )] = a;
				}
			}
		else if( s.kind == Stage::HbUp )
			{
			int t; const double * flt = hb_kernel( s.steep, s.third, t );
			for( int64_t o = 0; o < n_out; ++o )
				{
				const int64_t j = o >> 1;
				if( ( o & 1 ) == 0 ) { nxt[size_t( o )] = at( cur, j ); continue; }
				double a = flt[0] * ( at( cur, j + 1 ) + at( cur, j ) );
				for( int i = 1; i < t; ++i ) a += flt[i] * ( at( cur, j + 1 + i ) + at( cur, j - i ) );
				nxt[size_t( o )] = a;
				}
			}
		else if( s.kind == Stage::Conv )
			{
			const std::vector<double> & h = taps[k]; const int f2 = fl2[k]; const int64_t len_in = int64_t( cur.size() );
			for( int64_t o = 0; o < n_out; ++o )
				{
				const int64_t c = int64_t( s.down ) * o;
				int64_t m0 = c - f2 <= 0 ? 0 : ( c - f2 + s.up - 1 ) / s.up, m1 = ( c + f2 ) / s.up;
				if( m1 >= len_in ) m1 = len_in - 1;
				double a = 0.0;
				for( int64_t m = m0; m <= m1; ++m ) a += h[f2 + ( c - s.up * m )] * cur[size_t( m )];
				nxt[size_t( o )] = a;
				}
			}
		else if( s.whole )
			{
			const int fll = flt_len[k] / 2 - 1;
			for( int64_t o = 0; o < n_out; ++o )
				{
				const int64_t pos = o * s.in_step, p = pos / s.out_step;
				const double * ft = &taps[k][size_t( pos % s.out_step ) * flt_len[k]];
				double a = 0.0;
				for( int i = 0; i < flt_len[k]; ++i ) a += ft[i] * at( cur, p - fll + i );
				nxt[size_t( o )] = a;
				}
			}
		else
			{
			// CDSPFracInterpolator::process + convolve2, call by call (see oracle_resample_chain): what the stages in front have delivered
			// after `call` calls of `chunk` input samples each
			const int fll = flt_len[k] / 2 - 1, fl2i = flt_len[k] / 2;
			int64_t read_abs = 0, j = 0;
			int in_counter = 0, in_pos_int = 0;
			double shift = 0.0, frac = 0.0;
			for( int64_t call = 1; j < n_out; ++call )
				{
				int64_t w = call * chunk;
				for( int q = 0; q < k; ++q ) w = delivered( st[q], w, fl2[q] );
				while( w - read_abs > fl2i && j < n_out )
					{
					double x = frac * fracs[k];
					const int fti = int( x );
					x -= fti;
					const double x2 = x * x;
					const double * ftp = &taps[k][size_t( fti ) * flt_len[k] * 3];
					double a = 0.0;
					for( int i = 0; i < flt_len[k]; ++i ) a += ( ftp[3 * i] + ftp[3 * i + 1] * x + ftp[3 * i + 2] * x2 ) * at( cur, read_abs - fll + i );
					nxt[size_t( j++ )] = a;
					++in_counter;
					const double next = ( in_counter + shift ) * s.isrc / s.idst;
					const int next_int = int( next );
					read_abs += next_int - in_pos_int;
					in_pos_int = next_int;
					frac = next - next_int;
					}
				if( in_counter > 1000 ) { in_counter = 0; in_pos_int = 0; shift = frac * s.idst / s.isrc; }
				}
			}
		cur.swap( nxt );
		}
	for( int64_t k = 0; k < total_out; ++k ) out[k] = float( cur[size_t( k )] );
	return 0;
	}

} // namespace

// Audio::resample through whatever chain CDSPResampler builds for ( src, dst ); -1: a half-band chain deeper than the kernels restated
extern "C" int oracle_resample_general( const float * in, int64_t total_in, float * out, int64_t total_out, double src, double dst, int64_t chunk )
	{
	std::vector<Stage> st;
	if( !build_stages( src, dst, st ) ) return -1;
	return resample_by_stages( st, in, total_in, out, total_out, chunk );
	}

// the stage list as text for tests: "hbdown:1 conv:1/2 ..." -- returns the number of stages, 0 when not restated
extern "C" int oracle_resample_stage_list( double src, double dst, char * text, int capacity )
	{
	std::vector<Stage> st;
	if( !build_stages( src, dst, st ) ) return 0;
	std::string t;
	for( const Stage & s : st )
		{
		char b[96];
		if( s.kind == Stage::HbDown ) std::snprintf( b, sizeof b, "hbdown:%d%s ", s.steep, s.third ? "t" : "" );
		else if( s.kind == Stage::HbUp ) std::snprintf( b, sizeof b, "hbup:%d%s ", s.steep, s.third ? "t" : "" );
		else if( s.kind == Stage::Conv ) std::snprintf( b, sizeof b, "conv:%d/%d@%.6g,tb%.6g,g%.6g ", s.up, s.down, s.nf, s.tb, s.gain );
		else if( s.whole ) std::snprintf( b, sizeof b, "frac:%d/%d%s ", s.in_step, s.out_step, s.third ? "t" : "" );
		else std::snprintf( b, sizeof b, "spline:%.9g->%.9g%s ", s.isrc, s.idst, s.third ? "t" : "" );
		t += b;
		}
	if( text && capacity > 0 ) { std::snprintf( text, size_t( capacity ), "%s", t.c_str() ); }
	return int( st.size() );
	}

// 1 when this file restates the chain CDSPResampler( src, dst ) builds; the shape for tests
extern "C" int oracle_resample_chain_shape( double src, double dst, int * hb_down, int * up, int * down, double * norm_freq, double * gain, int * hb_up, int * third,
	int * interp, int * in_step, int * out_step, int * spline )
	{
	Chain ch;
	if( !chain_shape( src, dst, ch ) ) return 0;
	if( hb_down ) *hb_down = ch.hb_down;
	if( up ) *up = ch.up;
	if( down ) *down = ch.down;
	if( norm_freq ) *norm_freq = ch.norm_freq;
	if( gain ) *gain = ch.gain;
	if( hb_up ) *hb_up = ch.hb_up;
	if( third ) *third = ch.third ? 1 : 0;
	if( interp ) *interp = ch.interp ? 1 : 0;
	if( in_step ) *in_step = ch.in_step;
	if( out_step ) *out_step = ch.out_step;
	if( spline ) *spline = ch.spline ? 1 : 0;
	return 1;
	}

// Audio::resample through any restated chain: [half-band downsamplers] -> block convolver -> [half-band upsamplers] -> [interpolator].
// Every stage is evaluated on the whole stream in fp64 (r8brain's own intermediate type), zero before the start and past the input,
// for exactly as many samples as the next stage reads; the last stage rounds to float.
extern "C" int oracle_resample_chain( const float * in, int64_t total_in, float * out, int64_t total_out, double src, double dst, int64_t chunk )
	{
	Chain ch;
	if( !chain_shape( src, dst, ch ) ) return -1;
	if( total_out <= 0 ) return 0;
	std::vector<double> h; int fl2 = 0;
	if( !design_default_lowpass( ch.norm_freq, ch.gain, h, fl2 ) ) return -1;
	std::vector<double> bank; int flt_len = 0, fracs = 0;
	if( ch.interp && !ch.spline ) frac_delay_bank( ch.out_step, ch.third, bank, flt_len );
	if( ch.spline ) frac_spline_bank( ch.third, bank, flt_len, fracs );
	// how many samples of each stage's output the stage after it reads (backwards from the output)
	int64_t need = total_out;
	if( ch.interp && !ch.spline ) need = ( ( total_out - 1 ) * ch.in_step ) / ch.out_step - ( flt_len / 2 - 1 ) + flt_len;
	if( ch.spline ) need = int64_t( std::ceil( double( total_out ) * ch.isrc / ch.idst ) ) + flt_len + 8;
	std::vector<int64_t> need_up( size_t( ch.hb_up ) + 1 );
	need_up[size_t( ch.hb_up )] = need;                                            // output of the last upsampler (or of the convolver)
	for( int s = ch.hb_up - 1; s >= 0; --s ) { int n; hb_kernel( s, ch.third, n ); need_up[size_t( s )] = ( need_up[size_t( s ) + 1] - 1 ) / 2 + n + 1; }
	const int64_t need_conv = need_up[0];
	// forward
	std::vector<double> cur( in, in + total_in );
	for( int i = 0; i < ch.hb_down; ++i )                                          // SteepIndex c-1-i (CDSPResampler.h:358-365)
		{
		int n; const double * flt = hb_kernel( ch.hb_down - 1 - i, ch.third, n );
		std::vector<double> nxt( size_t( ( int64_t( cur.size() ) + 2 * n + 1 ) / 2 + 1 ) );   // the tail of the filter past the input's end
		for( int64_t j = 0; j < int64_t( nxt.size() ); ++j )
			{
			double sacc = at( cur, 2 * j );
			for( int t = 0; t < n; ++t ) sacc += flt[t] * ( at( cur, 2 * j + 2 * t + 1 ) + at( cur, 2 * j - 2 * t - 1 ) );
			nxt[size_t( j )] = sacc;
			}
		cur.swap( nxt );
		}
		{
		std::vector<double> nxt( size_t( need_conv > 0 ? need_conv : 0 ) );
		const int64_t len_in = int64_t( cur.size() );
		for( int64_t k = 0; k < need_conv; ++k )
			{
			const int64_t c = int64_t( ch.down ) * k;
			int64_t m0 = c - fl2 <= 0 ? 0 : ( c - fl2 + ch.up - 1 ) / ch.up;
			int64_t m1 = ( c + fl2 ) / ch.up;
			if( m1 >= len_in ) m1 = len_in - 1;
			double acc = 0.0;
			for( int64_t m = m0; m <= m1; ++m ) acc += h[fl2 + ( c - ch.up * m )] * cur[size_t( m )];
			nxt[size_t( k )] = acc;
			}
		cur.swap( nxt );
		}
	for( int s = 0; s < ch.hb_up; ++s )                                            // SteepIndex s (:203-209)
		{
		int n; const double * flt = hb_kernel( s, ch.third, n );
		std::vector<double> nxt( size_t( need_up[size_t( s ) + 1] ) );
		for( int64_t o = 0; o < int64_t( nxt.size() ); ++o )
			{
			const int64_t j = o >> 1;
			if( ( o & 1 ) == 0 ) { nxt[size_t( o )] = at( cur, j ); continue; }
			double sacc = flt[0] * ( at( cur, j + 1 ) + at( cur, j ) );
			for( int t = 1; t < n; ++t ) sacc += flt[t] * ( at( cur, j + 1 + t ) + at( cur, j - t ) );
			nxt[size_t( o )] = sacc;
			}
		cur.swap( nxt );
		}
	if( !ch.interp ) { for( int64_t k = 0; k < total_out; ++k ) out[k] = float( at( cur, k ) ); return 0; }
	const int fll = flt_len / 2 - 1;
	if( ch.spline )
		{
		// CDSPFracInterpolator::process + convolve2 (:826-895, :960-1005), call by call: oneshot() feeds `chunk` input samples per call
		// (CDSPResampler.h:494-552: the resampler was built with MaxInLen = the channel's frame count), of which the block convolver has
		// delivered max( 0, up * fed - latency ) samples of y; outputs come while more than flt_len / 2 samples lie ahead of the read position
		// (:964); the position counter is re-based after every call that leaves it above 1000 (:884-895)
		const int64_t latency = block_convolver_latency( fl2, ch.up );
		const int fl2i = flt_len / 2;
		int64_t read_abs = 0, j = 0;
		int in_counter = 0, in_pos_int = 0;
		double shift = 0.0, frac = 0.0;                                              // InitFracPos = 0 (linear-phase filters)
		for( int64_t call = 1; j < total_out; ++call )
			{
			const int64_t w = std::max<int64_t>( 0, call * chunk * ch.up - latency );
			while( w - read_abs > fl2i && j < total_out )
				{
				double x = frac * fracs;
				const int fti = int( x );
				x -= fti;
				const double x2 = x * x;
				const double * ftp = &bank[size_t( fti ) * flt_len * 3];
				double sacc = 0.0;
				for( int i = 0; i < flt_len; ++i ) sacc += ( ftp[3 * i] + ftp[3 * i + 1] * x + ftp[3 * i + 2] * x2 ) * at( cur, read_abs - fll + i );
				out[j++] = float( sacc );
				++in_counter;
				const double next = ( in_counter + shift ) * ch.isrc / ch.idst;
				const int next_int = int( next );
				read_abs += next_int - in_pos_int;
				in_pos_int = next_int;
				frac = next - next_int;
				}
			if( in_counter > 1000 ) { in_counter = 0; in_pos_int = 0; shift = frac * ch.idst / ch.isrc; }
			}
		return 0;
		}
	for( int64_t k = 0; k < total_out; ++k )
		{
		const int64_t pos = k * ch.in_step, p = pos / ch.out_step;
		const double * ft = &bank[size_t( pos % ch.out_step ) * flt_len];
		double sacc = 0.0;
		for( int i = 0; i < flt_len; ++i ) sacc += ft[i] * at( cur, p - fll + i );
		out[k] = float( sacc );
		}
	return 0;
	}

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
