// flan/flan.h -- umbrella header of the MI355X phase-vocoder path.
#pragma once
#include "flan/defines.h"
#include "flan/Function.h"
#include "flan/AudioBuffer.h"
#include "flan/PVBuffer.h"
#include "flan/Audio.h"
#include "flan/PV.h"
