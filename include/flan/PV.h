// flan/PV.h -- the PV side of the phase-vocoder path (mirrors the reference's src/flan/PV/PV.h:27-96,270-310,420-432:
// conversions and the frame processors named by BASELINE.json's configs).
#pragma once
#include <string>

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

	// ---- conversions ----
	/** Phase accumulation, inverse FFT, Hann window, overlap-add (Conversions/AudioPV.cpp:86-139).  A NaN/Inf in the data
	 *  prints the reference's warning and processing carries on. */
	Audio convert_to_audio( flan_CANCEL_ARG ) const;                             // PV.h:88-90
	Audio convert_to_lr_audio( flan_CANCEL_ARG ) const;                          // PV.h:94-96
	Audio convertToAudio( flan_CANCEL_ARG ) const;                               // older spelling

	// ---- frame processors ----
	PV modify_frequency( const Function<TF, Frequency> & mod, const Interpolator & = Interpolator::linear() ) const;  // PV.h:276-279
	PV modify_time( const Function<TF, Second> & mod, const Interpolator & = Interpolator::linear() ) const;          // PV.h:285-288
	PV repitch( const Function<TF, float> & factor, const Interpolator & = Interpolator::linear() ) const;            // PV.h:294-297
	PV stretch( const Function<TF, float> & factor, const Interpolator & = Interpolator::linear() ) const;            // PV.h:303-306
	PV shape( const Function<MF, MF> & shaper, bool use_shift_alignment = false ) const;                              // PV.h:426-429
	/** Extension (not in the reference): shape with the affine shaper mf -> { a*m + b, c*f + d } evaluated on the device;
	 *  identical to shape( [=]( MF mf ){ return MF{ a*mf.m + b, c*mf.f + d }; }, use_shift_alignment ). */
	PV shape_affine( float a, float b, float c, float d, bool use_shift_alignment = false ) const;
	};

} // namespace flan
