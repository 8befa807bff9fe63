// sub_launch.h -- the dft 512 / 256 several-chains-per-wavefront kernels (pv_kernels_sub.h) live in a translation unit of their own (sub.hip); this is
// what conversions.hip sees of them.
#pragma once
#include "flanhip_internal.h"
#include "pv_kernels.h"

namespace flanhip {

// dft 512 (32 lanes per chain) or 256 (16), hop = 1, 2, 4 or 8 steps of 2 LP samples, hop <= window, the window a multiple of a step
bool sub_shape( int dft, int W, int hop );
int sub_target_chains( int dft );
int sub_group_size( int dft );                     // chains per block = per group of the group totals the analysis leaves and the synthesis' carry prologue reads                  // chains the device holds at once
int run_analyze_sub( const AnalyzeParams & p, int dft, hipStream_t s );
int run_synth_sub( const SynthParams & p, int dft, hipStream_t s );

} // namespace flanhip
