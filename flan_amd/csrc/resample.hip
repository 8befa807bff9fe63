// resample.hip -- Audio::resample for the 2:1 decimation case behind the C ABI (BASELINE config 5: 96 kHz -> 48 kHz).
//
// Reference: Audio/AudioConversions.cpp:14-30 calls r8b::CDSPResampler( src, dst, num_frames ) with default parameters
// (transition band 2 %, attenuation 206.91 dB, linear phase; r8brain/CDSPResampler.h:115-118) and ONE oneshot over the whole
// channel-major buffer -- all channels as a single stream.  For src = 2 dst that is one low-pass (CDSPFIRFilter::buildLPFilter,
// r8brain/CDSPFIRFilter.h:227-493, normalised cut-off 1/2, gain 1) applied by FFT block convolution with 2:1 decimation and the
// filter latency consumed (r8brain/CDSPBlockConvolver.h:62-184).  Convolution is convolution: here it is the direct fp64 sum
//        out[k] = float( sum_{j=-fl2..fl2} h[j] * x[2k - j] ),   x = 0 outside the buffer,
// with the 1621 taps of r8brain's Kaiser-power windowed sinc design computed once on the host.
// The other ratios r8brain serves with a SINGLE block convolver (CDSPResampler.h:139-161: src*num == dst*den for (num,den) in
// (1,3) (2,3) (3,2) (3,4); :165-207: dst == 2 src, dst == 3 src) are the same sum over a zero-stuffed input,
//        out[k] = float( sum_j h[j] * xu[down*k - j] ),  xu[up*m] = x[m],
// with the low-pass at cut-off 1/max(up,down) and DC gain `up` (k_resample_rational).
// 44.1 <-> 48 kHz and the other ratios r8brain serves with one block convolver FOLLOWED BY one whole-stepping CDSPFracInterpolator
// (CDSPResampler.h:214-316 with no half-band stage, :319-378 likewise; CDSPFracInterpolator.h:573-602, :929-958) keep r8brain's two
// stages: k_resample_rational writes the band-limited stream y in fp64 (2x zero-stuffed, or filtered in place), k_frac_whole walks it
// with the bank of OutStep fractional-delay filters (28 taps at 206.91 dB).  Ratios that need half-band up/downsamplers (dst >= 2.02 src
// off the 2^k / 3*2^k grid, src >= 4 dst) or the spline-interpolated bank (no whole stepping: rates without a small common divisor)
// are not implemented: FLANHIP_ERR_UNSUPPORTED.
#include "flanhip_internal.h"
#include "processors_common.h"
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

// buildLPFilter for the parameters Flan always uses (tb 2 %, 206.91 dB, linear phase), cut-off `req_norm_freq`, DC gain `gain`
static bool design_default_lowpass( double req_norm_freq, double gain, std::vector<double> & h, int & fl2 )
	{
	const double tb = 2.0 * 0.01;                                              // CDSPFIRFilter.h:229
	double atten = -206.91 - 0.21;                                             // :233, :268-273
	const int corr_index = int( std::floor( ( -atten - 49.0 ) * 264 / 176.25 + 0.5 ) );   // :285-291
	if( corr_index != 237 ) return false;
	atten -= -19 / 196.0;                                                      // :354-376, entry 237 of the tb < 0.10 correction table
	const double pwr = 7.43932822146293e-8 * sqr( atten ) + 0.000102747434588003 * std::cos( 0.00785021930010397 * atten ) *
		std::cos( 0.633854318781239 + 0.103208573657699 * atten ) - 0.00798132247867036 - 0.000903555213543865 * atten -
		0.0969365532127236 * std::exp( 0.0779275237937911 * atten ) - 1.37304948662012e-5 * atten * std::cos( 0.00785021930010397 * atten );   // :379-384
	double hl, fo1;
	if( pwr <= 0.067665322581 )                                                // :386, :425-435
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

// Which single-step ratio is this?  CDSPResampler.h:139-161 (first match wins), then :165-207 with no half-band stage.
static bool rational_ratio( double src, double dst, int & up, int & down )
	{
	static const int common[5][2] = { { 1, 2 }, { 1, 3 }, { 2, 3 }, { 3, 2 }, { 3, 4 } };
	for( const auto & c : common )
		if( src * c[0] == dst * c[1] ) { up = c[0]; down = c[1]; return true; }
	if( src * 2 == dst ) { up = 2; down = 1; return true; }
	if( src * 3 == dst ) { up = 3; down = 1; return true; }
	return false;
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
// half-band 13 / 7 / 5 / 4 taps, third-band 9 / 6 / 5.  Deeper chains (32x ...) are not served.
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
	const double * src = nullptr; int n = 0;
	if( third ) { if( steep == 0 ) { src = t0; n = 9; } else if( steep == 1 ) { src = t1; n = 6; } else if( steep == 2 ) { src = t2; n = 5; } }
	else { if( steep == 0 ) { src = h0; n = 13; } else if( steep == 1 ) { src = h1; n = 7; } else if( steep == 2 ) { src = h2; n = 5; } else if( steep == 3 ) { src = h3; n = 4; } }
	if( !src ) return false;
	k.n = n;
	for( int i = 0; i < 13; ++i ) k.c[i] = i < n ? src[i] : 0.0;
	return true;
	}

// What CDSPResampler( src, dst ) builds, for the chains served here:
//   [ hb_down half-band downsamplers ] -> block convolver ( up, down, low-pass at norm_freq with DC gain `gain` ) -> [ hb_up half-band
//   upsamplers ] -> [ whole-stepping fractional interpolator ]
struct ChainShape { int hb_down = 0, up = 1, down = 1; double norm_freq = 0.5, gain = 1.0; int hb_up = 0; bool third = false, interp = false; int in_step = 0, out_step = 0;
	bool spline = false; double isrc = 0.0, idst = 0.0; };   // spline: no whole stepping -- the interpolator runs from isrc to idst with the spline-interpolated bank

static bool chain_shape( double src, double dst, ChainShape & ch )
	{
	if( src == dst ) return false;
	HbTaps probe;
	static const int common[5][2] = { { 1, 2 }, { 1, 3 }, { 2, 3 }, { 3, 2 }, { 3, 4 } };          // CDSPResampler.h:142-170
	for( const auto & c : common )
		if( src * c[0] == dst * c[1] ) { ch.up = c[0]; ch.down = c[1]; ch.norm_freq = 1.0 / std::max( c[0], c[1] ); ch.gain = c[0]; return true; }
	for( int i = 2; i <= 3; ++i )                                                 // :174-212: dst = i 2^c src
		for( int c = 0; src * ( i << c ) <= dst; ++c )
			if( src * ( i << c ) == dst )
				{
				ch.up = i; ch.norm_freq = 1.0 / i; ch.gain = i; ch.hb_up = c; ch.third = ( i == 3 );
				return c == 0 || hb_kernel( c - 1, ch.third, probe );
				}
	if( dst * 2 > src )                                                           // :214-316
		{
		const double thresh = src * 1.01;
		int c = 0, div = 1;
		while( !( dst < thresh * ( div * 2 ) ) ) { div *= 2; ++c; }                // :229-244
		int t1, t2;
		if( c == 1 && whole_stepping( src * 2.0, dst, t1, t2 ) ) c = 0;            // :266-276
		if( c > 0 ) return false;                                                  // intermediate interpolation with its own low-pass design: not served
		ch.up = 2; ch.norm_freq = dst > src ? 0.5 : 0.5 * dst / src; ch.gain = 2.0; ch.interp = true;
		if( !whole_stepping( src * 2.0, dst, ch.in_step, ch.out_step ) ) { ch.spline = true; ch.isrc = src * 2.0; ch.idst = dst; }
		return true;
		}
	double check = dst * 4.0;                                                     // :319-331
	int c = 0;
	while( check <= src ) { ++c; check *= 2.0; ch.gain *= 0.5; }
	const int div = 1 << c;
	ch.hb_down = c;
	int downf = 0;
	for( int d = 2; d <= 3; ++d ) if( dst * div * d == src ) { downf = d; break; }  // :340-349
	if( downf ) { ch.down = downf; ch.norm_freq = 1.0 / downf; ch.third = ( downf == 3 ); }
	else                                                                          // :351-356, :372-376
		{
		ch.norm_freq = dst * div / src; ch.third = ch.norm_freq * 3.0 <= 1.0; ch.interp = true;
		if( !whole_stepping( src, dst * div, ch.in_step, ch.out_step ) )
			{
			if( c > 0 ) return false;                                                // half-band stages in front of the spline bank: not served
			ch.spline = true; ch.isrc = src; ch.idst = dst * div;
			}
		}
	return c == 0 || hb_kernel( c - 1, ch.third, probe );
	}

// CDSPFracDelayFilterBank( OutStep, 1, 2, 206.91, third ) (CDSPFracInterpolator.h:64-121, window parameters :289-348 -- the rows that
// cover 206.91 dB): `fracs` filters of flt_len taps, row r delays by ( fracs - r ) / fracs samples.  Each is a Kaiser-power windowed
// sinc sampled at t + delay, t = -fl2 .. fl2 - 1 (CDSPSincFilterGen.h:184-193, :246-257, :432-517), normalised to unit DC gain.
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

static void spline_segments( const ChainShape & ch, int fl2_conv, int flt_len, int64_t chunk, int64_t total_out, std::vector<FracSegment> & segs )
	{
	const int kernel_len = 2 * fl2_conv + 1;
	int bits = 0; while( ( ( kernel_len - 1 ) >> bits ) != 0 ) ++bits;             // getBitOccupancy( KernelLen - 1 )
	const int64_t latency = int64_t( ( 2 << std::max( bits, 1 ) ) - ( ( kernel_len - 1 + ch.up - 1 ) / ch.up ) * ch.up ) + fl2_conv;   // InputLen + the filter's latency
	const int fl2i = flt_len / 2;
	const double src = ch.isrc, dst = ch.idst;
	FracSegment cur{ 0, 0, 0, 0.0, 0.0 };
	auto read_abs = [&]( int64_t n ) { return n == 0 ? cur.r0 : cur.r0 + int64_t( ( double( n ) + cur.shift ) * src / dst ); };
	int64_t j = 0;
	for( int64_t call = 1; j < total_out; ++call )
		{
		const int64_t w = std::max<int64_t>( 0, call * chunk * ch.up - latency );   // samples of y written so far
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

struct ChainPlan { double * d_h = nullptr; double * d_bank = nullptr; int fl2 = 0, flt_len = 0, fracs = 0; ChainShape shape; };
static std::map<std::tuple<int, double, double>, ChainPlan> g_chain_plans;   // per (device, src, dst)

// Device copies of the taps for one (up, down): h[0 .. 2 fl2] in natural order, and for the 2:1 kernel
// d_he[q] = h[2 fl2 - 2 q] (q = 0 .. fl2), d_ho[q] = h[2 fl2 - 1 - 2 q] (q = 0 .. fl2 - 1): the taps in the order it walks them
struct ResamplePlan { double * d_h = nullptr; double * d_he = nullptr; double * d_ho = nullptr; int fl2 = 0; };
static std::mutex g_rs_mutex;
static std::map<std::tuple<int, int, int>, ResamplePlan> g_rs_plans;   // per (device, up, down)

static int get_resample_plan( int up, int down, const ResamplePlan ** out )
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	std::lock_guard<std::mutex> lock( g_rs_mutex );
	const auto key = std::make_tuple( device, up, down );
	auto it = g_rs_plans.find( key );
	if( it != g_rs_plans.end() ) { *out = &it->second; return FLANHIP_OK; }
	std::vector<double> h; int fl2 = 0;
	FLANHIP_REQUIRE( design_default_lowpass( 1.0 / std::max( up, down ), double( up ), h, fl2 ), FLANHIP_ERR_UNSUPPORTED, "low-pass design outside the restated range" );
	std::vector<double> he( fl2 + 1 ), ho( fl2 + 1, 0.0 );
	for( int q = 0; q <= fl2; ++q ) he[q] = h[2 * fl2 - 2 * q];
	for( int q = 0; q < fl2; ++q ) ho[q] = h[2 * fl2 - 1 - 2 * q];
	ResamplePlan p; p.fl2 = fl2;
	const size_t nh = h.size();
	FLANHIP_CHECK( hipMalloc( &p.d_h, sizeof( double ) * ( nh + 2 * ( fl2 + 1 ) ) ) );
	p.d_he = p.d_h + nh;
	p.d_ho = p.d_he + ( fl2 + 1 );
	FLANHIP_CHECK( hipMemcpy( p.d_h, h.data(), sizeof( double ) * nh, hipMemcpyHostToDevice ) );
	FLANHIP_CHECK( hipMemcpy( p.d_he, he.data(), sizeof( double ) * ( fl2 + 1 ), hipMemcpyHostToDevice ) );
	FLANHIP_CHECK( hipMemcpy( p.d_ho, ho.data(), sizeof( double ) * ( fl2 + 1 ), hipMemcpyHostToDevice ) );
	*out = &g_rs_plans.emplace( key, p ).first->second;
	return FLANHIP_OK;
	}

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
struct rs_pair { double e, o; };

template<int RLO, int RHI>
__device__ __forceinline__ void rs_phase( const rs_pair * px, const double * __restrict__ he, const double * __restrict__ ho, int j0, int j1, double ( &acc )[RS_R] )
	{
	#pragma unroll 2
	for( int j = j0; j < j1; ++j )
		{
		const rs_pair v = px[j];
		#pragma unroll
		for( int r = RLO; r <= RHI; ++r )
			{
			acc[r] = __builtin_fma( he[j - 64 * r], v.e, acc[r] );
			acc[r] = __builtin_fma( ho[j - 64 * r], v.o, acc[r] );
			}
		}
	}

// j = fl2 + 64 S: the last tap (even m = 2 fl2) of output S, a full pair of taps for the outputs after it; then the 63 steps that follow
template<int S>
__device__ __forceinline__ void rs_ramp_down( const rs_pair * px, const double * __restrict__ he, const double * __restrict__ ho, int fl2, double ( &acc )[RS_R] )
	{
	const int j = fl2 + 64 * S;
	acc[S] = __builtin_fma( he[fl2], px[j].e, acc[S] );
	if constexpr( S + 1 < RS_R )
		{
		rs_phase<S + 1, RS_R - 1>( px, he, ho, j, j + 1, acc );
		rs_phase<S + 1, RS_R - 1>( px, he, ho, j + 1, j + 64, acc );
		rs_ramp_down<S + 1>( px, he, ho, fl2, acc );
		}
	}

template<int R>
__device__ __forceinline__ void rs_ramp_up( const rs_pair * px, const double * __restrict__ he, const double * __restrict__ ho, double ( &acc )[RS_R] )
	{
	if constexpr( R < RS_R - 1 )
		{
		rs_phase<0, R>( px, he, ho, 64 * R, 64 * ( R + 1 ), acc );
		rs_ramp_up<R + 1>( px, he, ho, acc );
		}
	}

__global__ __launch_bounds__( 64 * RS_WAVES ) void k_resample_2to1( const float * __restrict__ in, int64_t total_in, const double * __restrict__ he,
	const double * __restrict__ ho, int fl2, float * __restrict__ out, int64_t total_out )
	{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	double * s_x = reinterpret_cast<double*>( smem );                       // [2 * RS_BLOCK_OUT + 2 * fl2 + 2], local index i <-> input x0 + i
	const int64_t k0 = int64_t( blockIdx.x ) * RS_BLOCK_OUT;
	const int64_t x0 = 2 * k0 - fl2;                                        // first input sample the block touches
	const int span = 2 * RS_BLOCK_OUT + 2 * fl2 + 2;
	for( int i = threadIdx.x; i < span; i += 64 * RS_WAVES )
		{
		const int64_t a = x0 + i;
		s_x[i] = ( a >= 0 && a < total_in ) ? double( in[a] ) : 0.0;
		}
	__syncthreads();
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int t = RS_WAVE_OUT * wave + lane;                                // outputs k0 + t + 64 r, r < RS_R
	const rs_pair * px = reinterpret_cast<const rs_pair*>( s_x ) + t;
	double acc[RS_R];
	#pragma unroll
	for( int r = 0; r < RS_R; ++r ) acc[r] = 0.0;
	rs_ramp_up<0>( px, he, ho, acc );                                       // j in [0, 448): outputs join one by one
	rs_phase<0, RS_R - 1>( px, he, ho, 64 * ( RS_R - 1 ), fl2, acc );       // all eight in flight
	rs_ramp_down<0>( px, he, ho, fl2, acc );                                // j in [fl2, fl2 + 448]: outputs finish one by one
	#pragma unroll
	for( int r = 0; r < RS_R; ++r )
		{
		const int64_t k = k0 + t + 64 * r;
		if( k < total_out ) out[k] = float( acc[r] );
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
__global__ __launch_bounds__( FRAC_BLOCK ) void k_frac_whole( const double * __restrict__ y, int64_t ny, const double * __restrict__ bank, int flt_len,
	int in_step, int out_step, float * __restrict__ out, int64_t total_out )
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
	out[k] = float( acc );
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
__global__ __launch_bounds__( FRAC_BLOCK ) void k_frac_spline( const double * __restrict__ y, int64_t ny, const double * __restrict__ bank, int flt_len, int fracs,
	const FracSegment * __restrict__ segs, int num_segs, double isrc, double idst, float * __restrict__ out, int64_t total_out )
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
	out[j] = float( acc );
	}

static int get_chain_plan( double src, double dst, const ChainPlan ** out )
	{
	int device = 0;
	FLANHIP_CHECK( hipGetDevice( &device ) );
	std::lock_guard<std::mutex> lock( g_rs_mutex );
	const auto key = std::make_tuple( device, src, dst );
	auto it = g_chain_plans.find( key );
	if( it != g_chain_plans.end() ) { *out = &it->second; return FLANHIP_OK; }
	ChainPlan p;
	FLANHIP_REQUIRE( chain_shape( src, dst, p.shape ), FLANHIP_ERR_UNSUPPORTED, "ratio not served" );
	std::vector<double> h, bank;
	FLANHIP_REQUIRE( design_default_lowpass( p.shape.norm_freq, p.shape.gain, h, p.fl2 ), FLANHIP_ERR_UNSUPPORTED, "low-pass design outside the restated range" );
	if( p.shape.interp && !p.shape.spline ) frac_delay_bank( p.shape.out_step, p.shape.third, bank, p.flt_len );
	if( p.shape.spline ) frac_spline_bank( p.shape.third, bank, p.flt_len, p.fracs );
	FLANHIP_CHECK( hipMalloc( &p.d_h, sizeof( double ) * ( h.size() + bank.size() + 1 ) ) );
	p.d_bank = p.d_h + h.size();
	FLANHIP_CHECK( hipMemcpy( p.d_h, h.data(), sizeof( double ) * h.size(), hipMemcpyHostToDevice ) );
	if( !bank.empty() ) FLANHIP_CHECK( hipMemcpy( p.d_bank, bank.data(), sizeof( double ) * bank.size(), hipMemcpyHostToDevice ) );
	*out = &g_chain_plans.emplace( key, p ).first->second;
	return FLANHIP_OK;
	}

static size_t rational_lds( int fl2, int up, int down, int & span )
	{
	span = int( ( int64_t( down ) * ( RSG_BLOCK - 1 ) + 2 * fl2 ) / up + 2 );
	return sizeof( double ) * ( size_t( 2 * fl2 + 2 ) + size_t( span ) );
	}

template<typename InT, typename OutT>
static int launch_rational( const InT * d_in, int64_t n_in, const ChainPlan & plan, OutT * d_out, int64_t n_out, hipStream_t s )
	{
	int span = 0;
	const size_t lds = rational_lds( plan.fl2, plan.shape.up, plan.shape.down, span );
	FLANHIP_REQUIRE( lds <= 160 * 1024, FLANHIP_ERR_UNSUPPORTED, "filter too long for the LDS staging" );
	FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_resample_rational<InT, OutT> ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
	hipLaunchKernelGGL( ( k_resample_rational<InT, OutT> ), dim3( (unsigned) ( ( n_out + RSG_BLOCK - 1 ) / RSG_BLOCK ) ), dim3( RSG_BLOCK ), lds, s,
		d_in, n_in, plan.d_h, plan.fl2, plan.shape.up, plan.shape.down, span, d_out, n_out );
	return FLANHIP_OK;
	}

// A chain with more than the block convolver in it: every stage on the whole stream in fp64 (r8brain's intermediate type), each for exactly
// as many samples as the next one reads (zeros before the start and past the end, like the zeros oneshot() feeds); the last stage rounds to
// float.  Intermediate streams live in the stream's memory pool.
static int resample_chain_dev( const float * d_in, int64_t total_in, int64_t chunk, double src, double dst, float * d_out, int64_t total_out, hipStream_t s )
	{
	const ChainPlan * plan = nullptr;
	if( int rc = get_chain_plan( src, dst, &plan ) ) return rc;
	const ChainShape & ch = plan->shape;
	HbTaps down_taps[4], up_taps[4];
	FLANHIP_REQUIRE( ch.hb_down <= 4 && ch.hb_up <= 4, FLANHIP_ERR_UNSUPPORTED, "half-band chain too deep" );
	for( int i = 0; i < ch.hb_down; ++i ) FLANHIP_REQUIRE( hb_kernel( ch.hb_down - 1 - i, ch.third, down_taps[i] ), FLANHIP_ERR_UNSUPPORTED, "half-band chain too deep" );   // CDSPResampler.h:358-365
	for( int i = 0; i < ch.hb_up; ++i ) FLANHIP_REQUIRE( hb_kernel( i, ch.third, up_taps[i] ), FLANHIP_ERR_UNSUPPORTED, "half-band chain too deep" );                       // :203-209
	// samples each stage has to deliver (backwards from the output)
	int64_t need_up[5];
	need_up[ch.hb_up] = !ch.interp ? total_out
		: ch.spline ? int64_t( std::ceil( double( total_out ) * ch.isrc / ch.idst ) ) + plan->flt_len + 8
		: ( ( total_out - 1 ) * ch.in_step ) / ch.out_step - ( plan->flt_len / 2 - 1 ) + plan->flt_len;
	for( int i = ch.hb_up - 1; i >= 0; --i ) need_up[i] = ( need_up[i + 1] - 1 ) / 2 + up_taps[i].n + 1;
	const int64_t need_conv = std::max<int64_t>( need_up[0], 1 );
	retain_pool_memory();
	std::vector<void*> temps;
	auto temp = [&]( int64_t count, double ** out ) -> int
		{
		FLANHIP_CHECK( hipMallocAsync( reinterpret_cast<void**>( out ), sizeof( double ) * size_t( std::max<int64_t>( count, 1 ) ), s ) );
		temps.push_back( *out );
		return FLANHIP_OK;
		};
	int rc = FLANHIP_OK;
	const double * cur = nullptr; int64_t cur_len = total_in;                     // cur == nullptr: the stream is still the float input
	for( int i = 0; i < ch.hb_down && !rc; ++i )
		{
		const int64_t n_out = ( cur_len + 2 * down_taps[i].n + 1 ) / 2 + 1;       // up to the end of the filter's tail
		double * nxt = nullptr;
		if( ( rc = temp( n_out, &nxt ) ) ) break;
		const dim3 grid( (unsigned) ( ( n_out + HB_BLOCK - 1 ) / HB_BLOCK ) );
		if( cur ) hipLaunchKernelGGL( k_hb_down<double>, grid, dim3( HB_BLOCK ), 0, s, cur, cur_len, down_taps[i], nxt, n_out );
		else hipLaunchKernelGGL( k_hb_down<float>, grid, dim3( HB_BLOCK ), 0, s, d_in, cur_len, down_taps[i], nxt, n_out );
		cur = nxt; cur_len = n_out;
		}
	const bool conv_is_last = ch.hb_up == 0 && !ch.interp;
	if( !rc )
		{
		if( conv_is_last ) rc = cur ? launch_rational<double, float>( cur, cur_len, *plan, d_out, total_out, s ) : launch_rational<float, float>( d_in, cur_len, *plan, d_out, total_out, s );
		else
			{
			double * nxt = nullptr;
			if( !( rc = temp( need_conv, &nxt ) ) )
				{
				rc = cur ? launch_rational<double, double>( cur, cur_len, *plan, nxt, need_conv, s ) : launch_rational<float, double>( d_in, cur_len, *plan, nxt, need_conv, s );
				cur = nxt; cur_len = need_conv;
				}
			}
		}
	for( int i = 0; i < ch.hb_up && !rc; ++i )
		{
		const int64_t n_out = need_up[i + 1];
		const dim3 grid( (unsigned) ( ( n_out + HB_BLOCK - 1 ) / HB_BLOCK ) );
		if( i == ch.hb_up - 1 && !ch.interp ) hipLaunchKernelGGL( k_hb_up<float>, grid, dim3( HB_BLOCK ), 0, s, cur, cur_len, up_taps[i], d_out, n_out );
		else
			{
			double * nxt = nullptr;
			if( ( rc = temp( n_out, &nxt ) ) ) break;
			hipLaunchKernelGGL( k_hb_up<double>, grid, dim3( HB_BLOCK ), 0, s, cur, cur_len, up_taps[i], nxt, n_out );
			cur = nxt; cur_len = n_out;
			}
		}
	if( !rc && ch.spline )
		{
		std::vector<FracSegment> segs;
		spline_segments( ch, plan->fl2, plan->flt_len, chunk, total_out, segs );
		FracSegment * d_segs = nullptr;
		if( hipMallocAsync( reinterpret_cast<void**>( &d_segs ), sizeof( FracSegment ) * segs.size(), s ) != hipSuccess ) { set_error( "hipMallocAsync failed" ); rc = FLANHIP_ERR_HIP; }
		else
			{
			temps.push_back( d_segs );
			if( hipMemcpyAsync( d_segs, segs.data(), sizeof( FracSegment ) * segs.size(), hipMemcpyHostToDevice, s ) != hipSuccess ) { set_error( "hipMemcpyAsync failed" ); rc = FLANHIP_ERR_HIP; }
			else
				{
				(void) hipStreamSynchronize( s );                                   // the table is pageable host memory that dies with this call
				hipLaunchKernelGGL( k_frac_spline, dim3( (unsigned) ( ( total_out + FRAC_BLOCK - 1 ) / FRAC_BLOCK ) ), dim3( FRAC_BLOCK ), 0, s,
					cur, cur_len, plan->d_bank, plan->flt_len, plan->fracs, d_segs, int( segs.size() ), ch.isrc, ch.idst, d_out, total_out );
				}
			}
		}
	else if( !rc && ch.interp )
		hipLaunchKernelGGL( k_frac_whole, dim3( (unsigned) ( ( total_out + FRAC_BLOCK - 1 ) / FRAC_BLOCK ) ), dim3( FRAC_BLOCK ), 0, s,
			cur, cur_len, plan->d_bank, plan->flt_len, ch.in_step, ch.out_step, d_out, total_out );
	const hipError_t launched = hipGetLastError();
	for( void * t : temps ) (void) hipFreeAsync( t, s );
	if( rc ) return rc;
	FLANHIP_CHECK( launched );
	return FLANHIP_OK;
	}

} // namespace flanhip

using namespace flanhip;

static const char * const k_unsupported_ratio =
	"implemented: the single-step ratios (src:dst = 2:1, 3:1, 3:2, 2:3, 4:3, 1:2, 1:3), block convolver + interpolator ratios (any dst in "
	"( src/4, 2.02 src ): 44.1 <-> 48 kHz, 44.1 kHz -> 48001 Hz ...) and half-band chains (4x, 8x, 16x, 6x, 12x up; src >= 4 dst down); this "
	"one needs r8brain's intermediate interpolation with its own low-pass design, half-band stages in front of its spline-interpolated "
	"filter bank, or a deeper half-band chain";

extern "C" {

int64_t flanhip_resample_out_frames( int64_t num_frames, float src_rate, float dst_rate )
	{
	// AudioConversions.cpp:22  format.num_frames *= new_sample_rate / get_sample_rate()   (Frame *= float)
	return int64_t( int32_t( float( int32_t( num_frames ) ) * ( dst_rate / src_rate ) ) );
	}

int flanhip_resample_dev( const float * d_in, int64_t ch, int64_t n, float src_rate, float dst_rate, float * d_out, void * stream )
	{
	FLANHIP_REQUIRE( d_in && d_out && ch > 0 && n > 0 && src_rate > 0.0f && dst_rate > 0.0f, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	int up = 0, down = 0;
	const bool single = rational_ratio( double( src_rate ), double( dst_rate ), up, down );
	ChainShape shape;
	FLANHIP_REQUIRE( single || chain_shape( double( src_rate ), double( dst_rate ), shape ), FLANHIP_ERR_UNSUPPORTED, k_unsupported_ratio );
	if( int rc = require_device() ) return rc;
	const int64_t n_out = flanhip_resample_out_frames( n, src_rate, dst_rate );
	const int64_t total_in = ch * n, total_out = ch * n_out;
	if( total_out <= 0 ) return FLANHIP_OK;
	if( !single ) return resample_chain_dev( d_in, total_in, n, double( src_rate ), double( dst_rate ), d_out, total_out, (hipStream_t) stream );
	const ResamplePlan * plan = nullptr;
	if( int rc = get_resample_plan( up, down, &plan ) ) return rc;
	if( up == 1 && down == 2 && plan->fl2 >= 64 * ( RS_R - 1 ) )
		{
		const size_t lds = sizeof( double ) * size_t( 2 * RS_BLOCK_OUT + 2 * plan->fl2 + 2 );
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_resample_2to1 ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( k_resample_2to1, dim3( (unsigned) ( ( total_out + RS_BLOCK_OUT - 1 ) / RS_BLOCK_OUT ) ), dim3( 64 * RS_WAVES ), lds, (hipStream_t) stream,
			d_in, total_in, plan->d_he, plan->d_ho, plan->fl2, d_out, total_out );
		}
	else
		{
		int span = 0;
		const size_t lds = rational_lds( plan->fl2, up, down, span );
		FLANHIP_REQUIRE( lds <= 160 * 1024, FLANHIP_ERR_UNSUPPORTED, "filter too long for the LDS staging" );
		FLANHIP_CHECK( hipFuncSetAttribute( reinterpret_cast<const void*>( k_resample_rational<float, float> ), hipFuncAttributeMaxDynamicSharedMemorySize, int( lds ) ) );
		hipLaunchKernelGGL( ( k_resample_rational<float, float> ), dim3( (unsigned) ( ( total_out + RSG_BLOCK - 1 ) / RSG_BLOCK ) ), dim3( RSG_BLOCK ), lds, (hipStream_t) stream,
			d_in, total_in, plan->d_h, plan->fl2, up, down, span, d_out, total_out );
		}
	FLANHIP_CHECK( hipGetLastError() );
	return FLANHIP_OK;
	}

int flanhip_resample( const float * in, int64_t ch, int64_t n, float src_rate, float dst_rate, float * out, volatile int * cancel )
	{
	FLANHIP_REQUIRE( in && out && ch > 0 && n > 0, FLANHIP_ERR_INVALID_ARG, "bad arguments" );
	{ int up = 0, down = 0; ChainShape shape;
	  FLANHIP_REQUIRE( rational_ratio( double( src_rate ), double( dst_rate ), up, down ) || chain_shape( double( src_rate ), double( dst_rate ), shape ),
		FLANHIP_ERR_UNSUPPORTED, k_unsupported_ratio ); }
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
