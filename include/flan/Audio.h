// flan/Audio.h -- the Audio side of the phase-vocoder path (mirrors the reference's src/flan/Audio/Audio.h:25-176 for
// construction and conversions; every method is const and returns a fresh object, invalid input gives a null object).
#pragma once
#include <vector>

#include "flan/AudioBuffer.h"
#include "flan/defines.h"

namespace flan {

class PV;

class Audio : public AudioBuffer
	{
public:
	Audio();                                                                     // null Audio
	Audio( AudioBuffer && other );
	Audio copy() const;

	static Audio create_null();                                                  // prints "Null Audio created" (AudioConstructors.cpp:19-23)
	static Audio create_from_buffer( std::vector<float> && buffer, Channel num_channels, FrameRate sample_rate );   // Audio.h:62-66
	static Audio create_from_format( const AudioBuffer::Format & );
	static Audio create_empty_with_length( Second length, Channel num_channels = 1, FrameRate sample_rate = 48000.0f );
	static Audio create_empty_with_frames( Frame num_frames, Channel num_channels = 1, FrameRate sample_rate = 48000.0f ); // Audio.h:93-97

	// ---- conversions ----
	/** Windowed STFT + per-bin phase vocoding (Conversions/AudioPV.cpp:12-78).  dft_size: any EVEN size >= window_size up to 2^20 (the reference hands it
	 *  to FFTW as it is, FFTHelper.cpp:16-26).  Powers of two from 32 to 16384 run register / LDS FFT kernels (tuned ones at 256 ... 16384); sizes whose
	 *  half factors into 2 ... 13 a mixed-radix transform; sizes above 16384 whose half is C1 <= 256 times a product of 2 ... 13 up to 4096 (32768, 20000,
	 *  44100, 48000 ...) a two-level split; every other size up to 262144 (a large prime factor in the half) Bluestein's chirp-z form; only above that
	 *  what none of these serves falls to the direct sums (O( window x bins ) per frame, same results).
	 *  An odd size, or one below the window, returns a null PV. */
	PV convert_to_PV( Frame window_size = 2048, Frame hop = 128, Frame dft_size = 4096, flan_CANCEL_ARG ) const;  // Audio.h:158-163
	/** Stereo only: mid/side first (AudioPV.cpp:80-84). */
	PV convert_to_ms_PV( Frame window_size = 2048, Frame hop = 128, Frame dft_size = 4096, flan_CANCEL_ARG ) const;
	Audio convert_to_mid_side() const;                                           // AudioConversions.cpp:32-51
	Audio convert_to_left_right() const;                                         // :53-56
	/** r8brain-equivalent sample-rate conversion (AudioConversions.cpp:14-30). */
	Audio resample( FrameRate new_sample_rate ) const;

	// the older camelCase spellings BASELINE.json's north_star uses
	PV convertToPV( Frame window_size = 2048, Frame hop = 128, Frame dft_size = 4096, flan_CANCEL_ARG ) const;
	};

} // namespace flan
