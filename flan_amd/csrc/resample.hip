// resample.hip -- Audio::resample behind the C ABI: the whole of r8brain's CDSPResampler for the parameters Flan uses.
//
// Reference: Audio/AudioConversions.cpp:14-30 calls r8b::CDSPResampler( src, dst, num_frames ) with default parameters (transition band 2 %,
// attenuation 206.91 dB, linear phase; r8brain/CDSPResampler.h:115-118) and ONE oneshot over the whole channel-major buffer -- all channels as
// a single stream.  CDSPResampler's constructor (:119-378) turns the two rates into a chain of stages; build_stages() below restates it:
//   * block convolver (CDSPBlockConvolver.h, filter from CDSPFIRFilter::buildLPFilter :227-493): convolution is convolution -- here the direct
//     fp64 sum  out[k] = sum_m h[fl2 + down k - up m] x[m]  over the zero-stuffed input, latency consumed, taps of r8brain's Kaiser-power
//     windowed sinc design computed on the host.  Tuned kernels: k_resample_down<2 | 3> (BASELINE config 5's 96 -> 48 kHz is one such stage,
//     1621 taps), k_resample_up<1 | 2 | 3> (the 2x zero-stuffing convolver every interpolating chain starts with; the 1:1 low-pass in front of a
//     decimating interpolator); 2:3, 3:2, 3:4 and filters shorter than the kernels' ramps: k_resample_rational;
//   * half-band up / downsamplers (CDSPHBUpsampler.h, CDSPHBDownsampler.h): k_hb_up, k_hb_down with the kernels r8brain picks at 206.91 dB;
//   * fractional interpolator (CDSPFracInterpolator.h): whole stepping (44.1 <-> 48 kHz: a bank of OutStep fractional-delay filters, exact
//     integer positions: k_frac_whole) or, for rates without a small common divisor, the spline-interpolated bank with its per-call
//     position re-basing (k_frac_spline + spline_segments).
// Chains of more than one stage run stage by stage on the whole stream in fp64 (resample_stages_dev).  Nothing r8brain serves for these
// parameters is refused.
#include "flanhip_internal.h"
#include "processors_common.h"
#include "resample_fft.h"
#include <cmath>
#include <vector>
#include <tuple>
#include <algorithm>

namespace flanhip {

static double sqr( double x ) { return x * x; }

// Abramowitz & Stegun 9.8.1 / 9.8.2, the approximation of I0 r8brain windows with (r8bbase.h:1216-1236)
static double bessel_i0_as( double x )
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

// buildLPFilter (CDSPFIRFilter.h:227-493) for 206.91 dB, linear phase -- what Flan always asks for -- at cut-off `req_norm_freq`, DC gain `gain`
// and transition band `tb_percent` (2 everywhere but in the second convolver of r8brain's intermediate-interpolation branch)
static bool design_lowpass( double req_norm_freq, double tb_percent, double gain, std::vector<double> & h, int & fl2 )
	{
	const double tb = tb_percent * 0.01;                                       // :229
	if( !( tb > 0.0 ) ) return false;
	double atten = -206.91 - ( tb >= 0.25 ? 1.60 : tb >= 0.10 ? 0.69 : 0.21 );  // :233-281
	const int corr_index = int( std::floor( ( -atten - 49.0 ) * 264 / 176.25 + 0.5 ) );   // :283-291
	if( corr_index != ( tb >= 0.25 ? 239 : tb >= 0.10 ? 238 : 237 ) ) return false;
	atten -= tb >= 0.25 ? -12 / 101.0 : tb >= 0.10 ? -62 / 210.0 : -19 / 196.0;  // :293-376: the entry of each correction table this attenuation lands on
	const double pwr = 7.43932822146293e-8 * sqr( atten ) + 0.000102747434588003 * std::cos( 0.00785021930010397 * atten ) *
		std::cos( 0.633854318781239 + 0.103208573657699 * atten ) - 0.00798132247867036 - 0.000903555213543865 * atten -
		0.0969365532127236 * std::exp( 0.0779275237937911 * atten ) - 1.37304948662012e-5 * atten * std::cos( 0.00785021930010397 * atten );   // :379-384
	double hl, fo1;
	const bool low = pwr <= 0.067665322581;                                    // :386
	if( tb >= 0.25 && low )                                                    // :388-401
		{
		hl = 2.6778150875894 / tb + 300.547590563091 * std::atan( std::atan( 2.68959772209918 * pwr ) ) / ( 5.5099277187035 * tb - tb * std::tanh( std::cos( std::asinh( atten ) ) ) );
		fo1 = 0.987205355829873 * tb + 1.00011788929851 * std::atan2( -0.321432067051302 - 6.19131357321578 * std::sqrt( pwr ),
			hl + -1.14861472207245 / ( hl - 14.1821147585957 ) + std::pow( 0.9521145021664, std::pow( std::atan2( 1.12018764830637, tb ), 2.10988901686912 * hl - 20.9691278378345 ) ) );
		}
	else if( tb >= 0.10 && low )                                               // :403-414
		{
		hl = ( 1.56688617018066 + 142.064321294568 * pwr + 0.00419441117131136 * std::cos( 243.633511747297 * pwr ) - 0.022953443903576 * atten -
			0.026629568860284 * std::cos( 127.715550622571 * pwr ) ) / tb;
		fo1 = 0.982299356642411 * tb + 0.999441744774215 * std::asinh( ( -0.361783054039583 - 5.80540593623676 * std::sqrt( pwr ) ) / hl );
		}
	else if( low )                                                             // :416-426
		{
		hl = ( 2.45739657014937 + 269.183679500541 * pwr * std::cos( 5.73225668178813 +
			std::atan2( std::cosh( 0.988861169868941 - 17.2201556280744 * pwr ), 1.08340138240431 * pwr ) ) ) / tb;
		fo1 = 2.291956939 * tb + 0.01942450693 * sqr( tb ) * hl - 4.67538973161837 * pwr * tb - 1.668433124 * tb * std::pow( pwr, pwr );
		}
	else if( tb >= 0.25 )                                                      // :430-438
		{
		hl = ( 1.50258368698213 + 158.556968859477 * std::asinh( pwr ) * std::tanh( 57.9466246871383 * std::tanh( pwr ) ) - 0.0105440479814834 * atten ) / tb;
		fo1 = 0.994024401639321 * tb + ( -0.236282717577215 - 6.8724924545387 * std::sqrt( std::sin( pwr ) ) ) / hl;
		}
	else if( tb >= 0.10 )                                                      // :440-449
		{
		hl = ( 1.50277377248945 + 158.222625721046 * std::asinh( pwr ) * std::tanh( 1.02875299001715 + 42.072277322604 * pwr ) - 0.0108380943845632 * atten ) / tb;
		fo1 = 0.992539376734551 * tb + ( -0.251747813037178 - 6.74159892452584 * std::sqrt( std::tanh( std::tanh( std::tan( pwr ) ) ) ) ) / hl;
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
	const double beta = 125.0, kdiv = bessel_i0_as( beta );                    // CDSPFIRFilter.h:464, CDSPSincFilterGen.h:584-588
	h.assign( 2 * fl2 + 1, 0.0 );
	auto window = [&]( int t )                                                 // CDSPSincFilterGen.h:246-257, to the power pwr (:354)
		{
		const double n = 1.0 - sqr( t / len2 );
		return std::pow( n < 0.0 ? 0.0 : bessel_i0_as( beta * std::sqrt( n ) ) / kdiv, pwr );
		};
	h[fl2] = freq2 * window( 0 ) / pi;                                         // :353
	for( int t = 1; t <= fl2; ++t )                                            // :356-365
		{
		const double v = std::sin( freq2 * t ) * window( t ) / t / pi;
		h[fl2 + t] = v; h[fl2 - t] = v;
		}
	double s = 0.0;                                                            // CDSPFIRFilter.h:496-507
	for( double v : h ) s += v;
	for( double & v : h ) v *= gain / s;
	return true;
	}

// ---- two-stage ratios: block convolver + whole-stepping fractional interpolator ---------------------------------------------------
// CDSPFracInterpolator.h:539-558 (subtractive, at most 50 rounds) and :573-602
static bool whole_stepping( double src, double dst, int & in_step, int & out_step )
	{
	double l = src, s = dst, gcd = 0.0;
	bool found = false;
	for( int it = 0; it < 50 && !found; ++it )
		{
		if( s <= 0.0 ) { gcd = l; found = true; break; }
		const double r = l - s;
		l = s;
		s = std::fabs( r );
		}
	if( !found || gcd < 1.0 ) return false;
	const double i0 = src / gcd, o0 = dst / gcd;
	in_step = int( i0 ); out_step = int( o0 );
	return i0 == in_step && o0 == out_step && out_step <= 1500;
	}

// The half-band kernels getHBFilter / getHBFilterThird select at ReqAtten = 206.91 dB, by SteepIndex (CDSPHBUpsampler.h:43-215, :296-436):
// half-band 13 / 7 / 5 / 4 / 4 / 3 taps (SteepIndex 0, 1, 2, 3, 4-5, 6+), third-band 9 / 6 / 5 / 4 / 3 / 3.
struct HbTaps { double c[13]; int n; };
static bool hb_kernel( int steep, bool third, HbTaps & k )
	{
	static const double h0[13] = { 6.2816416238782957e-001, -1.8809076918442266e-001, 9.0918539368474965e-002, -4.6765502172995604e-002, 2.3287520069933797e-002,
		-1.0760626940880943e-002, 4.4853921118213676e-003, -1.6438774496992904e-003, 5.1441308429384374e-004, -1.3211724349740752e-004, 2.6191316362108199e-005,
		-3.5802424384280469e-006, 2.5491272423372411e-007 };
	static const double h1[7] = { 6.1610372237019151e-001, -1.5767891821295410e-001, 5.5089690570484962e-002, -1.6895755290596615e-002, 3.9416641999499014e-003,
		-6.0603620400878633e-004, 4.5632598748568398e-005 };
	static const double h2[5] = { 6.0626808278478261e-001, -1.3588224019070938e-001, 3.5544305138258458e-002, -6.5127022013993230e-003, 5.8255449020627736e-004 };
	static const double h3[4] = { 5.9835028661892165e-001, -1.1999986095168852e-001, 2.4132530901858028e-002, -2.4829565783680927e-003 };
	static const double t0[9] = { 6.2163188987470752e-001, -1.7108115412330563e-001, 6.9588371105224839e-002, -2.7339625869282957e-002, 9.2954473703765472e-003,
		-2.5537181861669997e-003, 5.2572296540671394e-004, -7.1813366796731157e-005, 4.8802392556669750e-006 };
	static const double t1[6] = { 6.1161456377889145e-001, -1.4743902036519768e-001, 4.5344160828746795e-002, -1.1207372108402218e-002, 1.8328498006058664e-003,
		-1.4518194076022933e-004 };
	static const double t2[5] = { 6.0590922849004858e-001, -1.3515953371903033e-001, 3.5020856634677522e-002, -6.3256195330255094e-003, 5.5506812768978109e-004 };
	static const double h45[4] = { 5.9819599535791312e-001, -1.1972157884617740e-001, 2.3977307400990484e-002, -2.4517239127622593e-003 };   // SteepIndex 4, 5
	static const double h6[3] = { 5.8594191093025305e-001, -9.7662866644414148e-002, 1.1720955714177778e-002 };                              // 6 and beyond
	static const double t3[4] = { 5.9823601283411165e-001, -1.1979369067338455e-001, 2.4017459011435899e-002, -2.4597811725236445e-003 };
	static const double t45[3] = { 5.8596887233874539e-001, -9.7703321108182931e-002, 1.1734448775437802e-002 };
	static const double t6[3] = { 5.8593945769687561e-001, -9.7659186594368730e-002, 1.1719728897494584e-002 };
	const double * src = nullptr; int n = 0;
	if( steep < 0 ) return false;
	if( third ) { if( steep == 0 ) { src = t0; n = 9; } else if( steep == 1 ) { src = t1; n = 6; } else if( steep == 2 ) { src = t2; n = 5; } else if( steep == 3 ) { src = t3; n = 4; }
		else if( steep <= 5 ) { src = t45; n = 3; } else { src = t6; n = 3; } }
	else { if( steep == 0 ) { src = h0; n = 13; } else if( steep == 1 ) { src = h1; n = 7; } else if( steep == 2 ) { src = h2; n = 5; } else if( steep == 3 ) { src = h3; n = 4; }
		else if( steep <= 5 ) { src = h45; n = 4; } else { src = h6; n = 3; } }
	if( !src ) return false;
	k.n = n;
	for( int i = 0; i < 13; ++i ) k.c[i] = i < n ? src[i] : 0.0;
	return true;
	}

// What CDSPResampler( src, dst ) builds (its whole constructor, CDSPResampler.h:119-378), as a list of stages.  Every stage is evaluated on the
// whole stream in fp64, r8brain's own intermediate type; the first reads the float audio, the last rounds to float.
struct Stage
	{
	enum Kind { HbDown, Conv, HbUp, Frac } kind = Conv;
	HbTaps taps{};                                                                // half-band stages
	int up = 1, down = 1; double nf = 0.5, tb = 2.0, gain = 1.0;                  // block convolver (tb in percent)
	bool third = false, whole = true; int in_step = 0, out_step = 0; double isrc = 0.0, idst = 0.0;   // interpolator (whole stepping, or the spline bank)
	const double * d_h = nullptr; int fl2 = 0;                                    // device tables, filled by get_stage_plan
	const double * d_poly = nullptr; int poly_len = 0, poly_u_min = 0;             // the taps as the tuned kernels walk them: `up` phases (down == 1), or `down` interleaves (up == 1)
	const double * d_bank = nullptr; int flt_len = 0, fracs = 0;
	const cd * d_spec = nullptr;                                                  // up == 1, down == 2: FFT( taps ) / 4096 for k_resample_ols2 (resample_fft.h)
	};

static bool build_stages( double src, double dst, std::vector<Stage> & st )
	{
	st.clear();
	if( !( src > 0.0 ) || !( dst > 0.0 ) || !( src <= 1e12 ) || !( dst <= 1e12 ) || src == dst ) return false;   // (r8brain asserts positive rates; NaN fails too)
	auto conv = [&]( int up, int down, double nf, double tb, double gain ) { Stage c; c.kind = Stage::Conv; c.up = up; c.down = down; c.nf = nf; c.tb = tb; c.gain = gain; st.push_back( c ); };
	auto hb = [&]( Stage::Kind k, int steep, bool third ) { Stage h; h.kind = k; h.third = third; const bool ok = hb_kernel( steep, third, h.taps ); st.push_back( h ); return ok; };
	auto frac = [&]( double isrc, double idst, bool third )
		{
		Stage f; f.kind = Stage::Frac; f.third = third; f.isrc = isrc; f.idst = idst;
		f.whole = whole_stepping( isrc, idst, f.in_step, f.out_step );
		st.push_back( f );
		};
	static const int common[5][2] = { { 1, 2 }, { 1, 3 }, { 2, 3 }, { 3, 2 }, { 3, 4 } };          // :142-170
	for( const auto & c : common )
		if( src * c[0] == dst * c[1] ) { conv( c[0], c[1], 1.0 / std::max( c[0], c[1] ), 2.0, c[0] ); return true; }
	for( int i = 2; i <= 3; ++i )                                                 // :174-212: dst = i 2^c src
		for( int c = 0; src * ( i << c ) <= dst; ++c )
			if( src * ( i << c ) == dst )
				{
				conv( i, 1, 1.0 / i, 2.0, i );
				for( int k = 0; k < c; ++k ) if( !hb( Stage::HbUp, k, i == 3 ) ) return false;
				return true;
				}
	if( dst * 2 > src )                                                           // :214-316
		{
		conv( 2, 1, dst > src ? 0.5 : 0.5 * dst / src, 2.0, 2.0 );                 // :218-225
		const double thresh = src * 1.01;
		int c = 0, div = 1;
		while( !( dst < thresh * ( div * 2 ) ) ) { div *= 2; ++c; }                // :229-244 (:246-261: the 3x variant tests 3 div against the threshold 2 div just failed: it never wins)
		int t1, t2;
		if( c == 1 && whole_stepping( src * 2.0, dst, t1, t2 ) ) c = 0;            // :266-276
		if( c > 0 )
			{
			// intermediate interpolation: to dst / 2^c first, then a 2x convolver whose transition band follows from the rates, then half-band upsamplers
			frac( src * 2.0 * div, dst, false );                                   // :293-294
			conv( 2, 1, 0.5, 100.0 * ( 1.0 - src * div / dst ) / 1.75, 2.0 );      // :296-302
			for( int i = 1; i < c; ++i ) if( !hb( Stage::HbUp, i - 1, false ) ) return false;   // :304-308
			}
		else frac( src * 2.0, dst, false );                                        // :312-313
		return true;
		}
	double check = dst * 4.0, fin_gain = 1.0;                                     // :319-331
	int c = 0;
	while( check <= src ) { ++c; check *= 2.0; fin_gain *= 0.5; }
	const int div = 1 << c;
	int downf = 1; double nf = 0.5; bool use_interp = true, third = false;
	for( int d = 2; d <= 3; ++d ) if( dst * div * d == src ) { downf = d; nf = 1.0 / d; use_interp = false; third = ( d == 3 ); break; }   // :340-349
	if( use_interp ) { nf = dst * div / src; third = nf * 3.0 <= 1.0; }           // :351-356
	for( int i = 0; i < c; ++i ) if( !hb( Stage::HbDown, c - 1 - i, third ) ) return false;   // :358-365
	conv( 1, downf, nf, 2.0, fin_gain );                                          // :367-370
	if( use_interp ) frac( src, dst * div, third );                               // :372-376
	return true;
	}

static void frac_window_params( bool third, double & beta, double & power, double & att, int & flt_len )
	{
	beta = third ? 19.1718281840114810 : 10.2382664677006100;
	power = third ? 1.2030083075440616 : 2.1608878780497056;
	att = third ? 215.2990 : 209.0609;                                            // the table row's own attenuation (what roundReqAtten turns 206.91 into)
	flt_len = third ? 22 : 28;
	}

// one fractional-delay filter: taps t = -fl2 .. fl2 - 1 of the windowed sinc at t + delay (generateFrac, CDSPSincFilterGen.h:432-517, for any
// delay the banks ask for: a negative one pushes the first tap out of the window, one above 1 the last), normalised to unit DC gain
static void frac_delay_row( int flt_len, double beta, double power, double delay, double * op )
	{
	const double pi = 3.14159265358979323846;
	const int fl2 = flt_len / 2;
	const double len2 = fl2, kdiv = bessel_i0_as( beta ), len2frac = delay / len2;
	const double f0 = std::sin( delay * pi );
	const bool unit = delay >= 1.0 - 1e-13 && delay <= 1.0 + 1e-13;
	double sum = 0.0;
	for( int i = 0; i < flt_len; ++i )
		{
		const int t = i - fl2;
		const double n = 1.0 - sqr( t / len2 + len2frac );
		const double w = std::pow( n < 0.0 ? 0.0 : bessel_i0_as( beta * std::sqrt( n ) ) / kdiv, power );
		const double ut = t + delay;
		double v;
		if( i == 0 && ut < -len2 ) v = 0.0;
		else if( t == ( unit ? -1 : 0 ) && std::fabs( ut ) <= 1e-13 ) v = w;     // the tap under the peak of the sinc
		else if( i == flt_len - 1 && ut > len2 ) v = 0.0;
		else v = ( ( t & 1 ) ? -f0 : f0 ) * w / ut / pi;
		op[i] = v;
		sum += v;
		}
	const double g = 1.0 / sum;
	for( int i = 0; i < flt_len; ++i ) op[i] *= g;
	}

static void frac_delay_bank( int fracs, bool third, std::vector<double> & bank, int & flt_len )
	{
	double beta, power, att;
	frac_window_params( third, beta, power, att, flt_len );
	bank.assign( size_t( fracs ) * flt_len, 0.0 );
	for( int r = 0; r < fracs; ++r ) frac_delay_row( flt_len, beta, power, double( fracs - r ) / fracs, &bank[size_t( r ) * flt_len] );
	}

// The bank of the interpolator WITHOUT whole stepping (getFilterBank( -1, 3, 8, ... ), CDSPFracInterpolator.h:97-100, :115-165): FilterFracs =
// ceil( 1.792462178761753 exp( 0.033300466782047 att ) ) rows -- att the table row's attenuation, 1893 / 2329 rows -- with the delays
// ( fracs - i ) / fracs, i = -3 .. fracs + 4; then every tap of rows 0 .. fracs becomes { x0, c1, c2 }, the 8-point 2nd-order spline through that
// tap of 8 consecutive rows (r8bbase.h:1019-1029): the kernel evaluates x0 + c1 x + c2 x^2 between two rows.
static void frac_spline_bank( bool third, std::vector<double> & bank, int & flt_len, int & fracs )
	{
	double beta, power, att;
	frac_window_params( third, beta, power, att, flt_len );
	fracs = int( std::ceil( 1.792462178761753 * std::exp( 0.033300466782047 * att ) ) );
	std::vector<double> raw( size_t( fracs + 8 ) * flt_len );
	for( int r = 0; r < fracs + 8; ++r ) frac_delay_row( flt_len, beta, power, double( fracs - ( r - 3 ) ) / fracs, &raw[size_t( r ) * flt_len] );
	bank.assign( size_t( fracs + 1 ) * flt_len * 3, 0.0 );
	for( int r = 0; r <= fracs; ++r )
		for( int i = 0; i < flt_len; ++i )
			{
			const double xm3 = raw[size_t( r ) * flt_len + i], xm2 = raw[size_t( r + 1 ) * flt_len + i], xm1 = raw[size_t( r + 2 ) * flt_len + i],
				x0 = raw[size_t( r + 3 ) * flt_len + i], x1 = raw[size_t( r + 4 ) * flt_len + i], x2 = raw[size_t( r + 5 ) * flt_len + i],
				x3 = raw[size_t( r + 6 ) * flt_len + i], x4 = raw[size_t( r + 7 ) * flt_len + i];
			double * c = &bank[( size_t( r ) * flt_len + i ) * 3];
			c[0] = x0;
			c[1] = ( 61.0 * ( x1 - xm1 ) + 16.0 * ( xm2 - x2 ) + 3.0 * ( x3 - xm3 ) ) / 76.0;
			c[2] = ( 106.0 * ( xm1 + x1 ) + 10.0 * x3 + 6.0 * xm3 - 3.0 * x4 - 29.0 * ( xm2 + x2 ) - 167.0 * x0 ) / 76.0;
			}
	}

// The interpolator without whole stepping re-bases its position counter at the end of every process() call that leaves it above 1000
// (CDSPFracInterpolator.h:884-895), and oneshot() makes one call per `chunk` input samples (CDSPResampler.h:494-552), of which the block
// convolver has delivered max( 0, up fed - latency ) samples of y (CDSPBlockConvolver.h:62-100, CDSPFIRFilter.h:467-478).  So the exact fp64
// read positions come in SEGMENTS: outputs j0 .. j0 + count - 1 read y at r0 + int( ( n + shift ) isrc / idst ), n = j - j0 (n = 0: r0 with
// fraction f0).  Worked out on the host call by call -- a bisection per call, not a walk over the outputs.
struct FracSegment { int64_t j0, count, r0; double shift, f0; };

// samples a stage has handed on once n samples have gone into it: it consumes its latency and then keeps pace with its input
// (CDSPHBDownsampler.h:95-150 with fl2 = 2 taps - 1; CDSPBlockConvolver.h:62-100, CDSPFIRFilter.h:467-478: InputLen + the filter's latency)
static int64_t delivered( const Stage & s, int64_t n )
	{
	if( s.kind == Stage::HbDown ) { const int fl2 = 2 * s.taps.n - 1; return n > fl2 ? ( n - fl2 + 1 ) >> 1 : 0; }
	if( s.kind == Stage::Conv )
		{
		const int kernel_len = 2 * s.fl2 + 1;
		int bits = 0; while( ( ( kernel_len - 1 ) >> bits ) != 0 ) ++bits;         // getBitOccupancy( KernelLen - 1 )
		const int64_t latency = int64_t( ( 2 << std::max( bits, 1 ) ) - ( ( kernel_len - 1 + s.up - 1 ) / s.up ) * s.up ) + s.fl2;
		return std::max<int64_t>( 0, n * s.up - latency ) / s.down;
		}
	return n;
	}

// the segments of interpolator stage k (see above): call c has seen c * chunk input samples of the whole chain
static void spline_segments( const std::vector<Stage> & st, int k, int64_t chunk, int64_t n_out, std::vector<FracSegment> & segs )
	{
	const Stage & f = st[size_t( k )];
	const int fl2i = f.flt_len / 2;
	const double src = f.isrc, dst = f.idst;
	FracSegment cur{ 0, 0, 0, 0.0, 0.0 };
	auto read_abs = [&]( int64_t n ) { return n == 0 ? cur.r0 : cur.r0 + int64_t( ( double( n ) + cur.shift ) * src / dst ); };
	int64_t j = 0;
	for( int64_t call = 1; j < n_out; ++call )
		{
		int64_t w = call * chunk;                                                  // samples of the stream in front of the interpolator written so far
		for( int q = 0; q < k; ++q ) w = delivered( st[size_t( q )], w );
		// outputs n = cur.count, cur.count + 1, ... come while w - read_abs( n ) > fl2i: the first n that fails, by bisection (read_abs is monotone)
		int64_t lo = cur.count, hi = cur.count;
		if( w - read_abs( lo ) > fl2i )
			{
			hi = lo + int64_t( double( w - fl2i - read_abs( lo ) ) * dst / src ) + 4;
			while( w - read_abs( hi ) > fl2i ) hi += 4;
			while( hi - lo > 1 ) { const int64_t mid = lo + ( hi - lo ) / 2; if( w - read_abs( mid ) > fl2i ) lo = mid; else hi = mid; }
			}
		j += hi - cur.count;
		cur.count = hi;
		if( cur.count > 1000 )                                                     // InCounter > 1000 at the end of the call: re-base
			{
			const double next = ( double( cur.count ) + cur.shift ) * src / dst;
			const int64_t next_int = int64_t( next );
			const double frac = next - double( next_int );
			segs.push_back( cur );
			cur = FracSegment{ j, 0, cur.r0 + next_int, frac * dst / src, frac };
			}
		}
	if( cur.count > 0 ) segs.push_back( cur );
	}

// device copies, cached for the process: low-pass taps per (device, cut-off, transition band, gain); interpolator banks per (device, rows or -1 for
// the spline bank, third-band); the stage list with its pointers per (device, src, dst)
struct DevTaps { double * d = nullptr; int fl2 = 0; double * d_poly = nullptr; int len = 0, u_min = 0; cd * d_spec = nullptr; };   // d: h[2 fl2 + 1]; poly: see Stage
struct DevBank { double * d = nullptr; int flt_len = 0, fracs = 0; };
static std::map<std::tuple<int, double, double, double, int, int>, DevTaps> g_dev_taps;
static std::map<std::tuple<int, int, bool>, DevBank> g_dev_banks;
static std::map<std::tuple<int, double, double>, std::vector<Stage>> g_stage_plans;

static std::mutex g_rs_mutex;

// out[k] = float( sum_m h[2 fl2 - m] x[2k - fl2 + m], m = 0 .. 2 fl2 ), one fp64 accumulator per output, m ascending (the checker's
// operation order).  1621 fp64 FMAs per output: the job of this kernel is to keep the fp64 pipes fed.
//   * A block stages the input span of its 2048 outputs in LDS as fp64, natural order.
//   * The pair P[j] = ( x[2(k0+t+j) - fl2], x[.. + 1] ) that output k0+t meets at taps m = 2j, 2j+1 is the pair output k0+t+64r meets
//     at m = 2(j-64r), 2(j-64r)+1.  So lane t owns the EIGHT outputs k0 + t + 64 r: one conflict-free 16-byte LDS read per j feeds
//     16 FMAs (a single output per thread needs 4x the LDS bandwidth a CU has).
//   * Taps are uniform across the wave: they arrive through the scalar cache as SGPR operands.
//   * Outputs enter and leave the j loop 64 steps apart: the loop is cut into phases with a compile-time set of active outputs, so
//     only real taps are ever multiplied (no zero padding: 0 * Inf must not leak into neighbours).
constexpr int RS_R = 8, RS_WAVES = 4, RS_WAVE_OUT = 64 * RS_R, RS_BLOCK_OUT = RS_WAVES * RS_WAVE_OUT;
template<int D> struct rs_group { double v[D]; };

// D: the decimation (2: the kernel described above; 3: 48 -> 16 kHz ...).  hd[d * ( Q + 1 ) + q] = h[2 fl2 - ( D q + d )], Q = 2 fl2 / D: the taps the
// D samples of group j meet; the last group is cut after d_last = 2 fl2 % D.
template<int D, int RLO, int RHI>
__device__ __forceinline__ void rs_phase( const rs_group<D> * px, const double * const ( &hd )[D], int j0, int j1, double ( &acc )[RS_R] )
	{
	#pragma unroll 2
	for( int j = j0; j < j1; ++j )
		{
		const rs_group<D> v = px[j];
		#pragma unroll
		for( int r = RLO; r <= RHI; ++r )
			{
			#pragma unroll
			for( int d = 0; d < D; ++d ) acc[r] = __builtin_fma( hd[d][j - 64 * r], v.v[d], acc[r] );
			}
		}
	}

// j = Q + 64 S: the last taps (d <= d_last) of output S, a full group of taps for the outputs after it; then the 63 steps that follow
template<int D, int S>
__device__ __forceinline__ void rs_ramp_down( const rs_group<D> * px, const double * const ( &hd )[D], int q1, int d_last, double ( &acc )[RS_R] )
	{
	const int j = q1 - 1 + 64 * S;
	const rs_group<D> v = px[j];
	#pragma unroll
	for( int d = 0; d < D; ++d ) if( d <= d_last ) acc[S] = __builtin_fma( hd[d][q1 - 1], v.v[d], acc[S] );
	if constexpr( S + 1 < RS_R )
		{
		if constexpr( D == 2 )
			{
			rs_phase<D, S + 1, RS_R - 1>( px, hd, j, j + 1, acc );             // (peeling the first step: the compiler's code for the rest is 6 % faster for D = 2,
			rs_phase<D, S + 1, RS_R - 1>( px, hd, j + 1, j + 64, acc );        //  25 % slower for D = 3 -- measured both ways)
			}
		else rs_phase<D, S + 1, RS_R - 1>( px, hd, j, j + 64, acc );
		rs_ramp_down<D, S + 1>( px, hd, q1, d_last, acc );
		}
	}

template<int D, int R>
__device__ __forceinline__ void rs_ramp_up( const rs_group<D> * px, const double * const ( &hd )[D], double ( &acc )[RS_R] )
	{
	if constexpr( R < RS_R - 1 )
		{
		rs_phase<D, 0, R>( px, hd, 64 * R, 64 * ( R + 1 ), acc );
		rs_ramp_up<D, R + 1>( px, hd, acc );
		}
	}

template<int D, typename InT, typename OutT>
__global__ __launch_bounds__( 64 * RS_WAVES ) void k_resample_down( const InT * __restrict__ in, int64_t total_in, const double * __restrict__ taps, int fl2,
	OutT * __restrict__ out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	double * s_x = reinterpret_cast<double*>( smem );                       // [D * RS_BLOCK_OUT + 2 * fl2 + D], local index i <-> input x0 + i
	const int64_t k0 = int64_t( blockIdx.x ) * RS_BLOCK_OUT;
	const int64_t x0 = D * k0 - fl2;                                        // first input sample the block touches
	const int span = D * RS_BLOCK_OUT + 2 * fl2 + D;
	for( int i = threadIdx.x; i < span; i += 64 * RS_WAVES )
		{
		const int64_t a = x0 + i;
		s_x[i] = ( a >= 0 && a < total_in ) ? double( in[a] ) : 0.0;
		}
	__syncthreads();
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int t = RS_WAVE_OUT * wave + lane;                                // outputs k0 + t + 64 r, r < RS_R
	const rs_group<D> * px = reinterpret_cast<const rs_group<D>*>( s_x ) + t;
	const int q1 = 2 * fl2 / D + 1, d_last = 2 * fl2 % D;
	const double * hd[D];
	#pragma unroll
	for( int d = 0; d < D; ++d ) hd[d] = taps + d * q1;
	double acc[RS_R];
	#pragma unroll
	for( int r = 0; r < RS_R; ++r ) acc[r] = 0.0;
	rs_ramp_up<D, 0>( px, hd, acc );                                        // j in [0, 448): outputs join one by one
	rs_phase<D, 0, RS_R - 1>( px, hd, 64 * ( RS_R - 1 ), q1 - 1, acc );     // all eight in flight (q1 - 1 >= 448: host check)
	rs_ramp_down<D, 0>( px, hd, q1, d_last, acc );                          // j in [Q, Q + 448]: outputs finish one by one
	#pragma unroll
	for( int r = 0; r < RS_R; ++r )
		{
		const int64_t k = k0 + t + 64 * r;
		if( k < total_out ) out[k] = OutT( acc[r] );
		}
	}

// The zero-stuffing convolver every interpolating chain starts with (2x; 3x for 16 -> 48 kHz ...), and the plain FIR of a 1:1 convolver stage (the
// low-pass in front of a decimating interpolator, 96 -> 44.1 kHz ...): y[PH n + p] = sum_m h[fl2 + PH n + p - PH m] x[m] is PH ordinary FIR filters
// over x, one per output phase -- g_p[i] = h[fl2 + p - PH ( u_min + i )] applied to x[n + u_min + i], all laid out over the union of their ranges
// (zero where a phase has no tap: for 2x each is 810 / 811 taps of 1621).  Same scheme as k_resample_down: a block stages the input span of its
// 2048 values of n in LDS as fp64; lane t owns n = n0 + t + 64 r, r < 8, whose PH outputs meet input x[n0 + t + u_min + j] at tap j - 64 r: one
// 8-byte LDS read per j feeds 8 PH FMAs; taps arrive as SGPR operands; ramp phases with compile-time active sets.
template<int PH, int RLO, int RHI>
__device__ __forceinline__ void up_phase( const double * px, const double * const ( &g )[PH], int j0, int j1, double ( &a )[PH][RS_R] )
	{
	#pragma unroll 2
	for( int j = j0; j < j1; ++j )
		{
		const double v = px[j];
		#pragma unroll
		for( int r = RLO; r <= RHI; ++r )
			{
			#pragma unroll
			for( int p = 0; p < PH; ++p ) a[p][r] = __builtin_fma( g[p][j - 64 * r], v, a[p][r] );
			}
		}
	}
template<int PH, int R> __device__ __forceinline__ void up_ramp_up( const double * px, const double * const ( &g )[PH], double ( &a )[PH][RS_R] )
	{
	if constexpr( R < RS_R - 1 ) { up_phase<PH, 0, R>( px, g, 64 * R, 64 * ( R + 1 ), a ); up_ramp_up<PH, R + 1>( px, g, a ); }
	}
template<int PH, int S> __device__ __forceinline__ void up_ramp_down( const double * px, const double * const ( &g )[PH], int len, double ( &a )[PH][RS_R] )
	{
	if constexpr( S < RS_R - 1 ) { up_phase<PH, S + 1, RS_R - 1>( px, g, len + 64 * S, len + 64 * ( S + 1 ), a ); up_ramp_down<PH, S + 1>( px, g, len, a ); }
	}

template<int PH, typename InT, typename OutT>
__global__ __launch_bounds__( 64 * RS_WAVES ) void k_resample_up( const InT * __restrict__ in, int64_t total_in, const double * __restrict__ taps,
	int len, int u_min, OutT * __restrict__ out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	double * s_x = reinterpret_cast<double*>( smem );                           // [RS_BLOCK_OUT + len], local index i <-> input n0 + u_min + i
	const int64_t n0 = int64_t( blockIdx.x ) * RS_BLOCK_OUT;
	const int span = RS_BLOCK_OUT + len;
	for( int i = threadIdx.x; i < span; i += 64 * RS_WAVES )
		{
		const int64_t a = n0 + u_min + i;
		s_x[i] = ( a >= 0 && a < total_in ) ? double( in[a] ) : 0.0;
		}
	__syncthreads();
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int t = RS_WAVE_OUT * wave + lane;                                    // n = n0 + t + 64 r
	const double * px = s_x + t;
	const double * g[PH];
	double a[PH][RS_R];
	#pragma unroll
	for( int p = 0; p < PH; ++p )
		{
		g[p] = taps + p * len;
		#pragma unroll
		for( int r = 0; r < RS_R; ++r ) a[p][r] = 0.0;
		}
	up_ramp_up<PH, 0>( px, g, a );                                              // j in [0, 448): the eight n join one by one
	up_phase<PH, 0, RS_R - 1>( px, g, 64 * ( RS_R - 1 ), len, a );              // all eight in flight (len >= 448: host check)
	up_ramp_down<PH, 0>( px, g, len, a );                                       // j in [len, len + 448): they finish one by one
	#pragma unroll
	for( int r = 0; r < RS_R; ++r )
		{
		const int64_t k = PH * ( n0 + t + 64 * r );
		#pragma unroll
		for( int p = 0; p < PH; ++p ) if( k + p < total_out ) out[k + p] = OutT( a[p][r] );
		}
	}

// The general single-step ratio: out[k] = float( sum_m h[fl2 + down*k - up*m] * x[m] ), m ascending over |down*k - up*m| <= fl2.
// One thread per output; a block of 256 outputs stages its input span (fp64) and the taps in LDS.
constexpr int RSG_BLOCK = 256;
__host__ __device__ inline int64_t rs_floor_div( int64_t a, int64_t b ) { return a >= 0 ? a / b : -( ( -a + b - 1 ) / b ); }
template<typename InT, typename OutT>   // float: the audio; double: the fp64 stream between the stages of a chain
__global__ __launch_bounds__( RSG_BLOCK ) void k_resample_rational( const InT * __restrict__ in, int64_t total_in, const double * __restrict__ taps, int fl2,
	int up, int down, int span, OutT * __restrict__ out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int ntaps = 2 * fl2 + 1;
	double * s_h = reinterpret_cast<double*>( smem );                       // [ntaps]
	double * s_x = s_h + ntaps + 1;                                         // [span], local index i <-> input m_base + i
	const int64_t k0 = int64_t( blockIdx.x ) * RSG_BLOCK;
	const int64_t m_base = rs_floor_div( int64_t( down ) * k0 - fl2 + up - 1, up );   // ceil( ( down*k0 - fl2 ) / up )
	for( int i = threadIdx.x; i < ntaps; i += RSG_BLOCK ) s_h[i] = taps[i];
	for( int i = threadIdx.x; i < span; i += RSG_BLOCK )
		{
		const int64_t m = m_base + i;
		s_x[i] = ( m >= 0 && m < total_in ) ? double( in[m] ) : 0.0;
		}
	__syncthreads();
	const int64_t k = k0 + threadIdx.x;
	if( k >= total_out ) return;
	const int64_t c = int64_t( down ) * k;
	const int64_t m_lo = rs_floor_div( c - fl2 + up - 1, up ), m_hi = rs_floor_div( c + fl2, up );
	const double * xp = s_x + ( m_lo - m_base );
	const double * hp = s_h + ( fl2 + c - int64_t( up ) * m_lo );           // tap of m_lo; the tap index falls by `up` per input sample
	const int count = int( m_hi - m_lo + 1 );
	double acc = 0.0;
	#pragma unroll 4
	for( int i = 0; i < count; ++i ) acc = __builtin_fma( hp[-up * i], xp[i], acc );   // zeros outside the buffer contribute +-0
	out[k] = OutT( acc );
	}

// CDSPFracInterpolator::convolve0 (CDSPFracInterpolator.h:929-958): output k sits k*in_step/out_step samples into y,
//        p = ( k*in_step ) / out_step,  r = ( k*in_step ) % out_step,  out[k] = float( sum_i bank[r][i] * y[p - ( flt_len/2 - 1 ) + i] ),
// i ascending, y = 0 before its start (the ring buffer's initial zeros, :771-778) and past what stage 1 wrote (never read: ny covers it).
constexpr int FRAC_BLOCK = 256;
template<typename OutT>
__global__ __launch_bounds__( FRAC_BLOCK ) void k_frac_whole( const double * __restrict__ y, int64_t ny, const double * __restrict__ bank, int flt_len,
	int in_step, int out_step, OutT * __restrict__ out, int64_t total_out )
	{
	const int64_t k = int64_t( blockIdx.x ) * FRAC_BLOCK + threadIdx.x;
	if( k >= total_out ) return;
	const int64_t pos = k * in_step, p = pos / out_step;
	const double * ft = bank + ( pos - p * out_step ) * flt_len;
	const int64_t a0 = p - ( flt_len / 2 - 1 );
	double acc = 0.0;
	if( a0 >= 0 && a0 + flt_len <= ny )
		for( int i = 0; i < flt_len; ++i ) acc = __builtin_fma( ft[i], y[a0 + i], acc );
	else
		for( int i = 0; i < flt_len; ++i )
			{
			const int64_t a = a0 + i;
			acc = __builtin_fma( ft[i], ( a >= 0 && a < ny ) ? y[a] : 0.0, acc );
			}
	out[k] = OutT( acc );
	}

// The same sums with the whole bank and the block's span of y in LDS (the bank rows of 64 neighbouring outputs are 64 different cache lines:
// gathered from memory, they are what k_frac_whole spends its time on).  A block owns 2048 consecutive outputs, a thread eight of them
// (eight independent fp64 chains).  Rows padded to flt_len + 1 doubles: the rows of a wavefront's lanes then start in different banks.
// Host checks: the bank fits (out_step <= ~400), in_step < 2^20 (32-bit positions inside a block).
constexpr int FW_OUT = 2048, FW_PER = FW_OUT / FRAC_BLOCK;
template<typename OutT>
__global__ __launch_bounds__( FRAC_BLOCK ) void k_frac_whole_lds( const double * __restrict__ y, int64_t ny, const double * __restrict__ bank, int flt_len,
	int in_step, int out_step, int span, OutT * __restrict__ out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int rs = flt_len + 1;
	double * s_bank = reinterpret_cast<double*>( smem );                       // [out_step][rs]
	double * s_y = s_bank + out_step * rs;                                      // [span], local index i <-> y[a_base + i]
	const int64_t k0 = int64_t( blockIdx.x ) * FW_OUT;
	const int64_t pos0 = k0 * in_step, p0 = pos0 / out_step;
	const unsigned r0 = unsigned( pos0 - p0 * out_step );
	const int64_t a_base = p0 - ( flt_len / 2 - 1 );
	for( int r = threadIdx.x / 32; r < out_step; r += FRAC_BLOCK / 32 )
		{
		const int i = threadIdx.x & 31;
		if( i < flt_len ) s_bank[r * rs + i] = bank[r * flt_len + i];
		}
	for( int i = threadIdx.x; i < span; i += FRAC_BLOCK )
		{
		const int64_t a = a_base + i;
		s_y[i] = ( a >= 0 && a < ny ) ? y[a] : 0.0;
		}
	__syncthreads();
	const double * ft[FW_PER]; const double * yp[FW_PER]; double acc[FW_PER];
	#pragma unroll
	for( int j = 0; j < FW_PER; ++j )
		{
		const unsigned q = r0 + unsigned( threadIdx.x + FRAC_BLOCK * j ) * unsigned( in_step );
		const unsigned dp = q / unsigned( out_step );
		ft[j] = s_bank + ( q - dp * unsigned( out_step ) ) * rs;
		yp[j] = s_y + dp;
		acc[j] = 0.0;
		}
	for( int i = 0; i < flt_len; ++i )
		{
		#pragma unroll
		for( int j = 0; j < FW_PER; ++j ) acc[j] = __builtin_fma( ft[j][i], yp[j][i], acc[j] );
		}
	#pragma unroll
	for( int j = 0; j < FW_PER; ++j )
		{
		const int64_t k = k0 + threadIdx.x + FRAC_BLOCK * j;
		if( k < total_out ) out[k] = OutT( acc[j] );
		}
	}

// CDSPHBDownsampler (CDSPHBDownsampler.h:95-260): out[j] = in[2j] + sum_t c[t] ( in[2j+2t+1] + in[2j-2t-1] ), in = 0 outside [0, n_in)
constexpr int HB_BLOCK = 256;
template<typename InT>
__global__ __launch_bounds__( HB_BLOCK ) void k_hb_down( const InT * __restrict__ in, int64_t n_in, HbTaps taps, double * __restrict__ out, int64_t n_out )
	{
	const int64_t j = int64_t( blockIdx.x ) * HB_BLOCK + threadIdx.x;
	if( j >= n_out ) return;
	auto at = [&]( int64_t i ) { return ( i >= 0 && i < n_in ) ? double( in[i] ) : 0.0; };
	double acc = at( 2 * j );
	for( int t = 0; t < taps.n; ++t ) acc = __builtin_fma( taps.c[t], at( 2 * j + 2 * t + 1 ) + at( 2 * j - 2 * t - 1 ), acc );
	out[j] = acc;
	}

// CDSPHBUpsampler (CDSPHBUpsampler.h:560-720): out[2j] = in[j], out[2j+1] = sum_t c[t] ( in[j+1+t] + in[j-t] )
template<typename OutT>
__global__ __launch_bounds__( HB_BLOCK ) void k_hb_up( const double * __restrict__ in, int64_t n_in, HbTaps taps, OutT * __restrict__ out, int64_t n_out )
	{
	const int64_t o = int64_t( blockIdx.x ) * HB_BLOCK + threadIdx.x;
	if( o >= n_out ) return;
	auto at = [&]( int64_t i ) { return ( i >= 0 && i < n_in ) ? in[i] : 0.0; };
	const int64_t j = o >> 1;
	double acc;
	if( ( o & 1 ) == 0 ) acc = at( j );
	else
		{
		acc = taps.c[0] * ( at( j + 1 ) + at( j ) );
		for( int t = 1; t < taps.n; ++t ) acc = __builtin_fma( taps.c[t], at( j + 1 + t ) + at( j - t ), acc );
		}
	out[o] = OutT( acc );
	}

// CDSPFracInterpolator::convolve2 (CDSPFracInterpolator.h:960-1005): output j of segment { j0, count, r0, shift, f0 } sits n = j - j0 steps in,
//        position = ( n + shift ) isrc / idst (fp64, in that order; n = 0: exactly r0 and f0),  p = r0 + int( position ),  frac = position - int( position );
//        x = frac * fracs, row = int( x ), x -= row;   out[j] = float( sum_i ( c0 + c1 x + c2 x^2 )[row][i] * y[p - ( flt_len/2 - 1 ) + i] )
template<typename OutT>
__global__ __launch_bounds__( FRAC_BLOCK ) void k_frac_spline( const double * __restrict__ y, int64_t ny, const double * __restrict__ bank, int flt_len, int fracs,
	const FracSegment * __restrict__ segs, int num_segs, double isrc, double idst, OutT * __restrict__ out, int64_t total_out )
	{
	const int64_t j = int64_t( blockIdx.x ) * FRAC_BLOCK + threadIdx.x;
	if( j >= total_out ) return;
	int lo = 0, hi = num_segs - 1;                                                // the last segment with j0 <= j
	while( lo < hi ) { const int mid = ( lo + hi + 1 ) >> 1; if( segs[mid].j0 <= j ) lo = mid; else hi = mid - 1; }
	const FracSegment sg = segs[lo];
	const int64_t n = j - sg.j0;
	int64_t p = sg.r0;
	double frac = sg.f0;
	if( n > 0 )
		{
		const double position = ( double( n ) + sg.shift ) * isrc / idst;
		const int64_t whole = int64_t( position );
		p += whole;
		frac = position - double( whole );
		}
	double x = frac * double( fracs );
	const int row = int( x );
	x -= double( row );
	const double x2 = x * x;
	const double * ft = bank + size_t( row ) * flt_len * 3;
	const int64_t a0 = p - ( flt_len / 2 - 1 );
	double acc = 0.0;
	for( int i = 0; i < flt_len; ++i )
		{
		const int64_t a = a0 + i;
		const double v = ( a >= 0 && a < ny ) ? y[a] : 0.0;
		acc = __builtin_fma( ( ft[3 * i] + ft[3 * i + 1] * x ) + ft[3 * i + 2] * x2, v, acc );
		}
	out[j] = OutT( acc );
	}

// ---- tables of k_resample_ols2 (resample_fft.h), worked out in long double ----------------------------------------------------------
static void ols_unit_circle( int n, std::vector<long double> & c, std::vector<long double> & sn )
	{
	c.resize( size_t( n ) ); sn.resize( size_t( n ) );
	const long double two_pi = 6.283185307179586476925286766559005768L;
	for( int m = 0; m < n; ++m ) { c[size_t( m )] = cosl( two_pi * m / n ); sn[size_t( m )] = sinl( two_pi * m / n ); }
	}

// spec[k] = ( 1 / N ) sum_t h[t] exp( -2 pi i k t / N ), k < N = 4096
static void ols_filter_spectrum( const std::vector<double> & h, std::vector<cd> & spec )
	{
	std::vector<long double> c, sn;
	ols_unit_circle( OLS_N, c, sn );
	spec.resize( OLS_N );
	for( int k = 0; k < OLS_N; ++k )
		{
		long double re = 0.0L, im = 0.0L;
		for( size_t t = 0; t < h.size(); ++t )
			{
			const size_t m = ( size_t( k ) * t ) & ( OLS_N - 1 );
			re += (long double) h[t] * c[m]; im -= (long double) h[t] * sn[m];
			}
		spec[size_t( k )] = cd{ double( re / OLS_N ), double( im / OLS_N ) };
		}
	}

static std::map<int, cd*> g_ols_twiddles;                                         // per device
static int ols_twiddles( const cd ** out )                                        // (caller holds g_rs_mutex)
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	auto it = g_ols_twiddles.find( device );
	if( it == g_ols_twiddles.end() )
		{
		std::vector<cd> tw( OlsTables::LEN + Ols3Tables::LEN );
		auto fill = [&]( int at, int ns, int period )                            // [15][ns]: exp( -2 pi i r k / period )
			{
			std::vector<long double> c, sn;
			ols_unit_circle( period, c, sn );
			for( int r = 1; r < 16; ++r )
				for( int k = 0; k < ns; ++k )
					{
					const size_t m = size_t( r * k ) % size_t( period );
					tw[size_t( at + ( r - 1 ) * ns + k )] = cd{ double( c[m] ), double( -sn[m] ) };
					}
			};
		fill( OlsTables::F1, 16, 256 ); fill( OlsTables::F2, 256, 4096 ); fill( OlsTables::I1, 8, 128 ); fill( OlsTables::I2, 128, 2048 );
		// k_resample_ols3's tables behind them: [7][ns]
		auto fill8 = [&]( int at, int ns, int period )
			{
			std::vector<long double> c, sn;
			ols_unit_circle( period, c, sn );
			for( int r = 1; r < 8; ++r )
				for( int k = 0; k < ns; ++k )
					{
					const size_t m = size_t( r * k ) % size_t( period );
					tw[size_t( OlsTables::LEN + at + ( r - 1 ) * ns + k )] = cd{ double( c[m] ), double( -sn[m] ) };
					}
			};
		fill8( Ols3Tables::F1, 8, 64 ); fill8( Ols3Tables::F2, 64, 512 ); fill8( Ols3Tables::F3, 512, 4096 );
		fill8( Ols3Tables::I1, 4, 32 ); fill8( Ols3Tables::I2, 32, 256 ); fill8( Ols3Tables::I3, 256, 2048 );
		cd * d = nullptr;
		FLANHIP_CHECK( hipMalloc( &d, sizeof( cd ) * tw.size() ) );
		FLANHIP_CHECK( hipMemcpy( d, tw.data(), sizeof( cd ) * tw.size(), hipMemcpyHostToDevice ) );
		it = g_ols_twiddles.emplace( device, d ).first;
		}
	*out = it->second;
	return FLANHIP_OK;
	}

// 0: always the direct sums (the checker's operation order); 1 (default): the FFT convolver where it applies

static int get_stage_plan( double src, double dst, const std::vector<Stage> ** out )
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	std::lock_guard<std::mutex> lock( g_rs_mutex );
	const auto key = std::make_tuple( device, src, dst );
	auto it = g_stage_plans.find( key );
	if( it != g_stage_plans.end() ) { *out = &it->second; return FLANHIP_OK; }
	std::vector<Stage> st;
	FLANHIP_REQUIRE( build_stages( src, dst, st ), FLANHIP_ERR_UNSUPPORTED, "ratio not served" );
	for( Stage & g : st )
		{
		if( g.kind == Stage::Conv )
			{
			const auto tkey = std::make_tuple( device, g.nf, g.tb, g.gain, g.up, g.down );
			auto tt = g_dev_taps.find( tkey );
			if( tt == g_dev_taps.end() )
				{
				std::vector<double> h, poly; DevTaps t;
				FLANHIP_REQUIRE( design_lowpass( g.nf, g.tb, g.gain, h, t.fl2 ), FLANHIP_ERR_UNSUPPORTED, "low-pass design outside the restated range" );
				const int fl2 = t.fl2;
				if( g.down == 1 && g.up <= 3 )
					{
					// the `up` output phases of a zero-stuffing convolver as FIR filters over the input (k_resample_up; up == 1: the filter itself, reversed):
					// phase p meets x[n + u] at tap fl2 + p - up u, 0 <= tap <= 2 fl2; all laid out over the union of their ranges of u
					t.u_min = -( fl2 / g.up ); t.len = ( fl2 + g.up - 1 ) / g.up - t.u_min + 1;
					poly.assign( size_t( g.up ) * t.len, 0.0 );
					for( int p2 = 0; p2 < g.up; ++p2 )
						for( int i = 0; i < t.len; ++i )
							{
							const int tap = fl2 + p2 - g.up * ( t.u_min + i );
							if( tap >= 0 && tap <= 2 * fl2 ) poly[size_t( p2 ) * t.len + i] = h[size_t( tap )];
							}
					}
				else if( g.up == 1 && g.down <= 3 )
					{
					// k_resample_down: interleave d of the reversed filter, poly[d * len + q] = h[2 fl2 - ( down q + d )]
					t.len = 2 * fl2 / g.down + 1;
					poly.assign( size_t( g.down ) * t.len, 0.0 );
					for( int m = 0; m <= 2 * fl2; ++m ) poly[size_t( m % g.down ) * t.len + m / g.down] = h[size_t( 2 * fl2 - m )];
					}
				const size_t nh = h.size();
				FLANHIP_CHECK( hipMalloc( &t.d, sizeof( double ) * ( nh + poly.size() ) ) );
				FLANHIP_CHECK( hipMemcpy( t.d, h.data(), sizeof( double ) * nh, hipMemcpyHostToDevice ) );
				if( !poly.empty() )
					{
					t.d_poly = t.d + nh;
					FLANHIP_CHECK( hipMemcpy( t.d_poly, poly.data(), sizeof( double ) * poly.size(), hipMemcpyHostToDevice ) );
					}
				if( g.up == 1 && g.down == 2 && 2 * fl2 + 1 <= OLS_N / 2 + 1 )
					{
					std::vector<cd> spec;
					ols_filter_spectrum( h, spec );
					// behind it the same spectrum in the order k_resample_ols3's forward transform leaves the bins in: [t][q] = H[rev( q ) + 512 t]
					spec.resize( 2 * OLS_N );
					for( int tq = 0; tq < OLS_N; ++tq )
						{
						const int t8 = tq >> 9, q = tq & 511, rev = ( ( q & 7 ) << 6 ) | ( q & 56 ) | ( q >> 6 );
						spec[size_t( OLS_N + tq )] = spec[size_t( rev + 512 * t8 )];
						}
					FLANHIP_CHECK( hipMalloc( &t.d_spec, sizeof( cd ) * spec.size() ) );
					FLANHIP_CHECK( hipMemcpy( t.d_spec, spec.data(), sizeof( cd ) * spec.size(), hipMemcpyHostToDevice ) );
					}
				tt = g_dev_taps.emplace( tkey, t ).first;
				}
			const DevTaps & dt = tt->second;
			g.d_h = dt.d; g.fl2 = dt.fl2; g.d_poly = dt.d_poly; g.poly_len = dt.len; g.poly_u_min = dt.u_min; g.d_spec = dt.d_spec;
			}
		else if( g.kind == Stage::Frac )
			{
			const auto bkey = std::make_tuple( device, g.whole ? g.out_step : -1, g.third );
			auto bt = g_dev_banks.find( bkey );
			if( bt == g_dev_banks.end() )
				{
				std::vector<double> bank; DevBank b;
				if( g.whole ) { frac_delay_bank( g.out_step, g.third, bank, b.flt_len ); b.fracs = g.out_step; }
				else frac_spline_bank( g.third, bank, b.flt_len, b.fracs );
				FLANHIP_CHECK( hipMalloc( &b.d, sizeof( double ) * bank.size() ) );
				FLANHIP_CHECK( hipMemcpy( b.d, bank.data(), sizeof( double ) * bank.size(), hipMemcpyHostToDevice ) );
				bt = g_dev_banks.emplace( bkey, b ).first;
				}
			g.d_bank = bt->second.d; g.flt_len = bt->second.flt_len; g.fracs = bt->second.fracs;
			}
		}
	*out = &g_stage_plans.emplace( key, std::move( st ) ).first->second;
	return FLANHIP_OK;
	}

static size_t rational_lds( int fl2, int up, int down, int & span )
	{
	span = int( ( int64_t( down ) * ( RSG_BLOCK - 1 ) + 2 * fl2 ) / up + 2 );
	return sizeof( double ) * ( size_t( 2 * fl2 + 2 ) + size_t( span ) );
	}

template<typename InT, typename OutT>
static int launch_rational( const InT * d_in, int64_t n_in, const Stage & g, OutT * d_out, int64_t n_out, hipStream_t s )
	{
	// 2:1 from float to float (BASELINE config 5's 96 -> 48 kHz and the like): overlap-save FFT convolution, like the reference's block convolver
	if constexpr( std::is_same<InT, float>::value && std::is_same<OutT, float>::value )
		{
		const int Lo = OLS_N / 2 - g.fl2;
		if( debug_options().resample_direct != 1 && g.d_spec && g.up == 1 && g.down == 2 && n_out >= 8 * int64_t( Lo ) )
			{
			const cd * tw = nullptr;
				{
				std::lock_guard<std::mutex> lock( g_rs_mutex );
				if( int rc = ols_twiddles( &tw ) ) return rc;
				}
			const int64_t blocks = ( n_out + 2 * int64_t( Lo ) - 1 ) / ( 2 * int64_t( Lo ) );
			if( debug_options().resample_direct == 2 )                              // (hook: the 256-thread radix-16 generation)
				{
				const size_t lds = sizeof( cd ) * OLS_BUF;
				FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_resample_ols2<InT, OutT> ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
				hipLaunchKernelGGL( ( k_resample_ols2<InT, OutT> ), dim3( (unsigned) blocks ), dim3( OLS_THREADS ), lds, s, d_in, n_in, g.d_spec, tw, g.fl2, d_out, n_out );
				return FLANHIP_OK;
				}
			const size_t lds = sizeof( cd ) * OLS3_BUF;
			FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_resample_ols3<InT, OutT> ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
			hipLaunchKernelGGL( ( k_resample_ols3<InT, OutT> ), dim3( (unsigned) blocks ), dim3( OLS3_THREADS ), lds, s, d_in, n_in, g.d_spec + OLS_N, tw + OlsTables::LEN, g.fl2, d_out, n_out );
			return FLANHIP_OK;
			}
		}
	// the tuned kernels (ramp phases of 64 steps per owned output: the filter must span them all)
	if( g.d_poly && g.poly_len - 1 >= 64 * ( RS_R - 1 ) )
		{
		const dim3 block( 64 * RS_WAVES );
		auto blocks = [&]( int64_t n ) { return dim3( (unsigned) ( ( n + RS_BLOCK_OUT - 1 ) / RS_BLOCK_OUT ) ); };
		if( g.down == 1 )
			{
			const size_t lds_up = sizeof( double ) * size_t( RS_BLOCK_OUT + g.poly_len );
			auto go = [&]( auto kernel )
				{
				FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kernel ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds_up ) ) );
				hipLaunchKernelGGL( kernel, blocks( ( n_out + g.up - 1 ) / g.up ), block, lds_up, s, d_in, n_in, g.d_poly, g.poly_len, g.poly_u_min, d_out, n_out );
				return FLANHIP_OK;
				};
			if( lds_up <= 160 * 1024 )
				{
				if( g.up == 1 ) return go( k_resample_up<1, InT, OutT> );
				if( g.up == 2 ) return go( k_resample_up<2, InT, OutT> );
				if( g.up == 3 ) return go( k_resample_up<3, InT, OutT> );
				}
			}
		else if( g.up == 1 )
			{
			const size_t lds_down = sizeof( double ) * size_t( g.down * RS_BLOCK_OUT + 2 * g.fl2 + g.down );
			auto go = [&]( auto kernel )
				{
				FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( kernel ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds_down ) ) );
				hipLaunchKernelGGL( kernel, blocks( n_out ), block, lds_down, s, d_in, n_in, g.d_poly, g.fl2, d_out, n_out );
				return FLANHIP_OK;
				};
			if( lds_down <= 160 * 1024 )
				{
				if( g.down == 2 ) return go( k_resample_down<2, InT, OutT> );
				if( g.down == 3 ) return go( k_resample_down<3, InT, OutT> );
				}
			}
		}
	int span = 0;
	const size_t lds = rational_lds( g.fl2, g.up, g.down, span );
	FLANHIP_REQUIRE( lds <= 160 * 1024, FLANHIP_ERR_UNSUPPORTED, "filter too long for the LDS staging" );
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_resample_rational<InT, OutT> ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	hipLaunchKernelGGL( ( k_resample_rational<InT, OutT> ), dim3( (unsigned) ( ( n_out + RSG_BLOCK - 1 ) / RSG_BLOCK ) ), dim3( RSG_BLOCK ), lds, s,
		d_in, n_in, g.d_h, g.fl2, g.up, g.down, span, d_out, n_out );
	return FLANHIP_OK;
	}

template<typename OutT>
static int launch_frac_whole( const double * d_y, int64_t ny, const Stage & g, OutT * d_out, int64_t n_out, hipStream_t s )
	{
	// the last output of a block sits ( r0 + 2047 in_step ) / out_step samples behind the first, r0 < out_step
	const int64_t span = ( int64_t( g.out_step - 1 ) + int64_t( FW_OUT - 1 ) * g.in_step ) / g.out_step + g.flt_len + 1;
	const size_t lds = sizeof( double ) * ( size_t( g.out_step ) * ( g.flt_len + 1 ) + size_t( span ) );
	if( g.flt_len <= 32 && g.in_step < ( 1 << 20 ) && lds <= 160 * 1024 )
		{
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_frac_whole_lds<OutT> ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( k_frac_whole_lds<OutT>, dim3( (unsigned) ( ( n_out + FW_OUT - 1 ) / FW_OUT ) ), dim3( FRAC_BLOCK ), lds, s,
			d_y, ny, g.d_bank, g.flt_len, g.in_step, g.out_step, int( span ), d_out, n_out );
		}
	else hipLaunchKernelGGL( k_frac_whole<OutT>, dim3( (unsigned) ( ( n_out + FRAC_BLOCK - 1 ) / FRAC_BLOCK ) ), dim3( FRAC_BLOCK ), 0, s,
		d_y, ny, g.d_bank, g.flt_len, g.in_step, g.out_step, d_out, n_out );
	return FLANHIP_OK;
	}

// A chain of more than one stage: every stage for exactly as many samples as the next one reads (zeros before the start and past the end, like
// the zeros oneshot() feeds), intermediate fp64 streams from the stream's memory pool.  `chunk`: the input samples per process() call of the
// reference (the channel's frame count) -- it decides where the spline interpolator re-bases its position counter.
static int resample_stages_dev( const float * d_in, int64_t total_in, int64_t chunk, double src, double dst, float * d_out, int64_t total_out, hipStream_t s )
	{
	const std::vector<Stage> * plan = nullptr;
	if( int rc = get_stage_plan( src, dst, &plan ) ) return rc;
	const std::vector<Stage> & st = *plan;
	const int ns = int( st.size() );
	FLANHIP_REQUIRE( ns > 0 && st[size_t( ns - 1 )].kind != Stage::HbDown, FLANHIP_ERR_UNSUPPORTED, "a chain cannot end in a half-band downsampler" );
	std::vector<int64_t> need( size_t( ns ), 0 );
	need[size_t( ns - 1 )] = total_out;
	for( int k = ns - 1; k > 0; --k )                                             // what stage k reads of stage k - 1
		{
		const Stage & g = st[size_t( k )];
		const int64_t n = need[size_t( k )];
		int64_t r;
		if( g.kind == Stage::HbDown ) r = 2 * ( n - 1 ) + 2 * g.taps.n;
		else if( g.kind == Stage::HbUp ) r = ( n - 1 ) / 2 + g.taps.n + 1;
		else if( g.kind == Stage::Conv ) r = ( int64_t( g.down ) * ( n - 1 ) + g.fl2 ) / g.up + 1;
		else if( g.whole ) r = ( ( n - 1 ) * g.in_step ) / g.out_step - ( g.flt_len / 2 - 1 ) + g.flt_len;
		else r = int64_t( std::ceil( double( n ) * g.isrc / g.idst ) ) + g.flt_len + 8;
		need[size_t( k - 1 )] = std::max<int64_t>( r, 1 );
		}
	retain_pool_memory();
	std::vector<void*> temps;
	auto temp = [&]( size_t bytes, void ** out ) -> int               // (an error here leaves the loop below; what was allocated before is freed after it)
		{
		FLANHIP_CHECK( hipMallocAsync( out, std::max<size_t>( bytes, 8 ), s ) );
		temps.push_back( *out );
		return FLANHIP_OK;
		};
	int rc = FLANHIP_OK;
	const double * cur = nullptr; int64_t cur_len = total_in;                     // cur == nullptr: the stream is still the float input
	for( int k = 0; k < ns && !rc; ++k )
		{
		const Stage & g = st[size_t( k )];
		const bool last = k == ns - 1;
		const int64_t n_out = need[size_t( k )];
		double * nxt = nullptr;
		if( !last ) { void * t = nullptr; if( ( rc = temp( sizeof( double ) * size_t( n_out ), &t ) ) ) break; nxt = static_cast<double*>( t ); }
		if( g.kind == Stage::HbDown )
			{
			const dim3 grid( (unsigned) ( ( n_out + HB_BLOCK - 1 ) / HB_BLOCK ) );
			if( cur ) hipLaunchKernelGGL( k_hb_down<double>, grid, dim3( HB_BLOCK ), 0, s, cur, cur_len, g.taps, nxt, n_out );
			else hipLaunchKernelGGL( k_hb_down<float>, grid, dim3( HB_BLOCK ), 0, s, d_in, cur_len, g.taps, nxt, n_out );
			}
		else if( g.kind == Stage::HbUp )
			{
			const dim3 grid( (unsigned) ( ( n_out + HB_BLOCK - 1 ) / HB_BLOCK ) );
			if( last ) hipLaunchKernelGGL( k_hb_up<float>, grid, dim3( HB_BLOCK ), 0, s, cur, cur_len, g.taps, d_out, n_out );
			else hipLaunchKernelGGL( k_hb_up<double>, grid, dim3( HB_BLOCK ), 0, s, cur, cur_len, g.taps, nxt, n_out );
			}
		else if( g.kind == Stage::Conv )
			{
			if( last ) rc = cur ? launch_rational<double, float>( cur, cur_len, g, d_out, n_out, s ) : launch_rational<float, float>( d_in, cur_len, g, d_out, n_out, s );
			else rc = cur ? launch_rational<double, double>( cur, cur_len, g, nxt, n_out, s ) : launch_rational<float, double>( d_in, cur_len, g, nxt, n_out, s );
			}
		else
			{
			const dim3 grid( (unsigned) ( ( n_out + FRAC_BLOCK - 1 ) / FRAC_BLOCK ) );
			if( g.whole ) rc = last ? launch_frac_whole<float>( cur, cur_len, g, d_out, n_out, s ) : launch_frac_whole<double>( cur, cur_len, g, nxt, n_out, s );
			else
				{
				std::vector<FracSegment> segs;
				spline_segments( st, k, chunk, n_out, segs );
				void * t = nullptr;
				if( ( rc = temp( sizeof( FracSegment ) * segs.size(), &t ) ) ) break;
				FracSegment * d_segs = static_cast<FracSegment*>( t );
				if( hipMemcpyAsync( d_segs, segs.data(), sizeof( FracSegment ) * segs.size(), hipMemcpyHostToDevice, s ) != hipSuccess ) { set_error( "hipMemcpyAsync failed" ); rc = FLANHIP_ERR_HIP; break; }
				(void) hipStreamSynchronize( s );                                   // the table is pageable host memory that dies with this call
				if( last ) hipLaunchKernelGGL( k_frac_spline<float>, grid, dim3( FRAC_BLOCK ), 0, s, cur, cur_len, g.d_bank, g.flt_len, g.fracs, d_segs, int( segs.size() ), g.isrc, g.idst, d_out, n_out );
				else hipLaunchKernelGGL( k_frac_spline<double>, grid, dim3( FRAC_BLOCK ), 0, s, cur, cur_len, g.d_bank, g.flt_len, g.fracs, d_segs, int( segs.size() ), g.isrc, g.idst, nxt, n_out );
				}
			}
		cur = nxt; cur_len = n_out;
		}
	const hipError_t launched = hipGetLastError();
	for( void * t : temps ) (void) hipFreeAsync( t, s );
	if( rc ) return rc;
	FLANHIP_CHECK( launched );
	return FLANHIP_OK;
	}

} // namespace flanhip

using namespace flanhip;

static const char * const k_unsupported_ratio =
	"no chain of r8brain's stages for these rates (equal rates are a copy, not a resampling)";

extern "C" {

int64_t flanhip_resample_out_frames( int64_t num_frames, float src_rate, float dst_rate )
	{
	// AudioConversions.cpp:22  format.num_frames *= new_sample_rate / get_sample_rate()   (Frame *= float)
	return int64_t( int32_t( float( int32_t( num_frames ) ) * ( dst_rate / src_rate ) ) );
	}


int flanhip_resample_dev( const float * d_in, int64_t ch, int64_t n, float src_rate, float dst_rate, float * d_out, void * stream )
	{
	FLANHIP_REQUIRE( d_in && d_out && ch > 0 && n > 0 && src_rate > 0.0f && dst_rate > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	std::vector<Stage> probe;
	FLANHIP_REQUIRE( build_stages( double( src_rate ), double( dst_rate ), probe ), FLANHIP_ERR_UNSUPPORTED, k_unsupported_ratio );
	if( int rc = require_device() ) return rc;
	const int64_t n_out = flanhip_resample_out_frames( n, src_rate, dst_rate );
	const int64_t total_in = ch * n, total_out = ch * n_out;
	if( total_out <= 0 ) return FLANHIP_OK;
	return resample_stages_dev( d_in, total_in, n, double( src_rate ), double( dst_rate ), d_out, total_out, (hipStream_t) stream );
	}

int flanhip_resample( const float * in, int64_t ch, int64_t n, float src_rate, float dst_rate, float * out, volatile int * cancel )
	{
	FLANHIP_REQUIRE( in && out && ch > 0 && n > 0 && src_rate > 0.0f && dst_rate > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	{ std::vector<Stage> probe;
	  FLANHIP_REQUIRE( build_stages( double( src_rate ), double( dst_rate ), probe ), FLANHIP_ERR_UNSUPPORTED, k_unsupported_ratio ); }
	if( int rc = require_device() ) return rc;
	if( cancelled( cancel ) ) return FLANHIP_ERR_CANCELLED;
	const int64_t n_out = flanhip_resample_out_frames( n, src_rate, dst_rate );
	float * d_in = nullptr; float * d_out = nullptr;
	FLANHIP_CHECK( hipMalloc( &d_in, sizeof( float ) * size_t( ch * n ) ) );
	if( hipMalloc( &d_out, sizeof( float ) * size_t( std::max<int64_t>( ch * n_out, 1 ) ) ) != hipSuccess ) { (void) hipFree( d_in ); set_error( "hipMalloc failed" ); return FLANHIP_ERR_HIP; }
	int rc = FLANHIP_OK;
	rc = flanhip_upload( d_in, in, sizeof( float ) * size_t( ch * n ) );
	if( !rc ) rc = flanhip_resample_dev( d_in, ch, n, src_rate, dst_rate, d_out, nullptr );
	if( !rc && hipDeviceSynchronize() != hipSuccess ) { set_error( "resample kernel failed" ); rc = FLANHIP_ERR_HIP; }
	if( !rc && cancelled( cancel ) ) rc = FLANHIP_ERR_CANCELLED;
	if( !rc ) rc = flanhip_download( out, d_out, sizeof( float ) * size_t( ch * n_out ) );
	(void) hipFree( d_in ); (void) hipFree( d_out );
	return rc;
	}

} // extern "C"
