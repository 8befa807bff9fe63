// flan/defines.h -- scalar types of the phase-vocoder path (mirrors the reference's src/flan/defines.h:10-62).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>

namespace flan {

// defines.h:10-29 of the reference, by kind: indices and counts are 32-bit integers, everything measured is a float
using Index = int;
using Channel = int32_t;    using Frame = int32_t;      using Bin = int32_t;        using Harmonic = int32_t;
using fFrame = float;       using fBin = float;         // fractional frame / bin positions
using Second = float;       using Sample = float;       using Frequency = float;    using Magnitude = float;
using FrameRate = float;    using Radian = float;

struct MF { Magnitude m; Frequency f; };   // defines.h:31-35
struct TF { Second t; Frequency f; };      // defines.h:37-41

const Radian pi = std::acos( -1.0f );       // defines.h:44 (a float)
const Radian pi2 = pi * 2.0f;               // defines.h:45

// defines.h:49-62: voluntary cancellation points.  Every long-running method takes `std::atomic<bool> & canceller`
// (defaulted to a flag that never changes) and returns a null object once it reads true.
inline std::atomic<bool> & default_canceller() { static std::atomic<bool> flag( false ); return flag; }
#define flan_CANCEL_ARG std::atomic<bool> & canceller = ::flan::default_canceller()
#define flan_CANCEL_ARG_CPP std::atomic<bool> & canceller

// Function.h:60-63 / Utility/execution.h:19-25: how a user callable may be evaluated when it is sampled on the host.
enum class ExecutionPolicy { Linear_Sequenced, Linear_Unsequenced, Parallel_Sequenced, Parallel_Unsequenced };

} // namespace flan
