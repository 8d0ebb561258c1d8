// Host-callable launchers implemented next to the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "lld_params.hpp"

namespace smilehip {
// frames per work tile of the kernel that will serve this geometry
int launch_tile_frames(int Nfft, int N, int force_generic);
bool fast512_applicable(int Nfft, int N);
hipError_t launch_mfcc(const LldParams &P, int force_generic, hipStream_t s);
hipError_t launch_delta(const DeltaParams &P, hipStream_t s);
}  // namespace smilehip
