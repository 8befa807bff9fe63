// team_launch.h -- the dft 8192 / 16384 team kernels (pv_kernels_team.h) live in a translation unit of their own (team.hip: they compile beside
// conversions.hip, not behind it); this is what conversions.hip sees of them.
#pragma once
#include "flanhip_internal.h"
#include "pv_kernels.h"

namespace flanhip {

// Does this shape run the team kernels?  dft 8192 (R = 4) or 16384 (R = 8), window and hop multiples of 128 R samples, window / 128 R one of 4, 8, 16
// and hop / 128 R one of the steps instantiated for it (team.hip).  Everything else at these sizes keeps its round-1 / mixed-radix kernels.
bool team_shape( int dft, int W, int hop );
int team_group_size( int dft );                    // chains (teams) per block: 2 at dft 8192, 1 at dft 16384
int team_target_chains( int dft );                 // chains the device holds at once: one block per chain, two blocks (R = 4) or one (R = 8) per CU
int run_analyze_team( const AnalyzeParams & p, const Plan & plan, int dft, hipStream_t s );
int run_synth_team( const SynthParams & p, const Plan & plan, int dft, hipStream_t s );

} // namespace flanhip
