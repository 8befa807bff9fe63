// flan/PV.h -- the PV side of the phase-vocoder path (mirrors the reference's src/flan/PV/PV.h:27-96,270-310,420-432:
// conversions and the frame processors named by BASELINE.json's configs).
#pragma once
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "flan/Function.h"
#include "flan/PVBuffer.h"
#include "flan/defines.h"

namespace flan {

class Audio;

class PV : public PVBuffer
	{
public:
	PV();                                                                        // 0-size PV with the default Format (PV.h:52)
	PV( PVBuffer && other );

	static PV create_null();
	static PV create_from_format( const PVBuffer::Format & );                    // PV.h:66-69
	static PV load_from_file( const std::string & filename );                    // PV.h:75-78
	PV copy() const;

	/** Function.h:155-171 through PV.h:31-35: sample f on this PV's (frame, bin) grid: argument TF{ frame/analysis_rate, bin_to_frequency(bin) } */
	template<typename T>
	FunctionSample2d<T> sample_function_over_domain( const Function<TF, T> & f ) const
		{
		return f.sample( 0, float( get_num_frames() ), 1.0f / get_analysis_rate(), 0, float( get_num_bins() ), bin_to_frequency( 1 ) );
		}

	/** PV.h:37-49: a function of time sampled at every frame */
	template<typename T>
	std::vector<T> sample_function_over_time_domain( const Function<Second, T> & f ) const
		{
		std::vector<T> out( static_cast<size_t>( get_num_frames() ) );
		detail::for_each_index( 0, get_num_frames(), f.get_execution_policy(), [&]( int frame ){ out[size_t( frame )] = f( frame_to_time( fFrame( frame ) ) ); } );
		return out;
		}

	/** A weighted approximation from the surrounding MFs (PV.h:203-222, PV.cpp:41-90); host-side accessors like get_MF */
	MF getBinInterpolated( Channel channel, fFrame frame, fBin bin, const Interpolator & interp = Interpolator::linear() ) const;
	MF getBinInterpolated( Channel channel, fFrame frame, Bin bin, const Interpolator & interp = Interpolator::linear() ) const;
	MF getBinInterpolated( Channel channel, Frame frame, fBin bin, const Interpolator & interp = Interpolator::linear() ) const;

	// ---- conversions ----
	/** Phase accumulation, inverse FFT, Hann window, overlap-add (Conversions/AudioPV.cpp:86-139).  A NaN/Inf in the data
	 *  prints the reference's warning and processing carries on. */
	Audio convert_to_audio( flan_CANCEL_ARG ) const;                             // PV.h:88-90
	Audio convert_to_lr_audio( flan_CANCEL_ARG ) const;                          // PV.h:94-96
	Audio convertToAudio( flan_CANCEL_ARG ) const;                               // older spelling

	// ---- frame processors ----
	/** The general time / frequency warp: every input quad is mapped by `mod` and rasterised into the output (PV.h:258-270).
	 *  Named interpolators only.  Outputs longer than ten minutes are refused like the reference refuses them. */
	PV modify( const Function<TF, TF> & mod, const Interpolator & = Interpolator::linear() ) const;                    // PV.h:267-270 (PVModify.cpp:15-193)
	PV modify_frequency( const Function<TF, Frequency> & mod, const Interpolator & = Interpolator::linear() ) const;  // PV.h:276-279
	PV modify_time( const Function<TF, Second> & mod, const Interpolator & = Interpolator::linear() ) const;          // PV.h:285-288
	PV repitch( const Function<TF, float> & factor, const Interpolator & = Interpolator::linear() ) const;            // PV.h:294-297
	PV stretch( const Function<TF, float> & factor, const Interpolator & = Interpolator::linear() ) const;            // PV.h:303-306
	PV shape( const Function<MF, MF> & shaper, bool use_shift_alignment = false ) const;                              // PV.h:426-429
	/** Extension (not in the reference): shape with the affine shaper mf -> { a*m + b, c*f + d } evaluated on the device;
	 *  identical to shape( [=]( MF mf ){ return MF{ a*mf.m + b, c*mf.f + d }; }, use_shift_alignment ). */
	PV shape_affine( float a, float b, float c, float d, bool use_shift_alignment = false ) const;

	// ---- further frame processors (all on the device; user functions are sampled on the host like the reference does) ----
	PV replace_amplitudes( const PV & amp_source, const Function<TF, float> & amount = 1 ) const;                     // PV.h:404-407 (PV.cpp:205-236)
	PV subtract_amplitudes( const PV & amp_source, const Function<TF, float> & amount = 1 ) const;                    // PV.h:414-417 (PV.cpp:238-264)
	PV retain_n_loudest_partials( const Function<Second, Bin> & num_bins ) const;                                     // PV.h:446-448 (PV.cpp:592-596)
	PV remove_n_loudest_partials( const Function<Second, Bin> & num_bins ) const;                                     // PV.h:455-457 (PV.cpp:598-602)
	PV resonate( Second length, const Function<TF, float> & decay ) const;                                            // PV.h:466-469 (PV.cpp:604-641)
	/** Only the named interpolators (Interpolator::linear() ... sine()) run here: the mix is evaluated per output MF on the device. */
	PV desample( const Function<TF, float> & decimation_ratio, const Interpolator & = Interpolator::linear() ) const; // PV.h:323-326 (PVModify.cpp:445-511)
	/** Any interpolator (it is sampled on the host, PVModify.cpp:631-633).  The reference reads frame time_to_frame(end_time)
	 *  unchecked (:649; one past the end for the default end_time = -1): here the end frame is clamped to the last frame. */
	PV time_extrapolate( Second start_time, Second end_time, Second extrapolation_time,
		const Interpolator & = Interpolator::linear() ) const;                                                        // PV.h:352-357 (PVModify.cpp:607-666)

	// ---- selecting, rearranging and re-placing frames and bins ----
	/** One frame, the linear blend of the two around `time` (clamped into the PV). */
	PV get_frame( Second time ) const;                                                                                // PV.h:199-201 (PV.cpp:24-39)
	/** Every output point reads the input point the selector names.  The selector is sampled on the host over the OUTPUT's grid. */
	PV select( Second length, const Function<TF, TF> & selector ) const;                                              // PV.h:236-239 (PV.cpp:92-127)
	/** Time freeze.  Of several pauses on one frame the first given is kept (unspecified in the reference: its sort is not stable). */
	PV freeze( const std::vector<Second> & pause_times, const std::vector<Second> & pause_lengths ) const;            // PV.h:247-250 (PV.cpp:129-198)
	/** Cubic-spline stretch: frame k of the input is followed by max( uint32( interpolation( time of k ) ), 1 ) output frames; every
	 *  bin's magnitudes and frequencies are splined through those knots in double precision (PV.h:308-316).  Needs 3+ frames. */
	PV stretch_spline( const Function<Second, float> & interpolation ) const;                                        // PV.h:314-316 (PVModify.cpp:387-443)
	/** Every MF becomes the distribution-weighted average of its bin over the surrounding smear_size seconds, sampled every
	 *  `granularity` frames (PV.h:327-340).  The default distribution is the reference's raised cosine. */
	PV smear_time( const Function<TF, Second> & smear_size, const Function<TF, int> & granularity = 5,
		const Function<Second, float> & distribution = []( Second t ){ return float( 0.5f * ( 1.0f + std::cos( 3.14159265358979323846 * t ) ) ); } ) const;   // (PVModify.cpp:513-605)
	PV add_octaves( const Function<std::pair<Second, Harmonic>, float> & series_scale ) const;                        // PV.h:387-389 (PV.cpp:409-413)
	PV add_harmonics( const Function<std::pair<Second, Harmonic>, float> & series_scale ) const;                      // PV.h:394-396 (PV.cpp:415-419)
	PV cut_frames( Frame start, Frame end ) const;                                                                    // PV.h:473-476 (PV.cpp:643-668)
	std::vector<PV> split_at_times( std::vector<Second> split_times ) const;                                          // PV.h:478-480 (PV.cpp:670-696)
	static PV join( const std::vector<const PV *> & ins );                                                            // PV.h:482-484 (PV.cpp:698-720)
	static PV join( const std::vector<PV> & ins );                                                                    // PV.h:486-488 (PV.cpp:722-727)
	};

} // namespace flan
