// Host-callable launchers implemented next to the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "lld_params.hpp"
#include "tables.hpp"

namespace smilehip {
struct Fast512Host {
  std::vector<float2> tw256, tw512;
  std::vector<uint2> mel_entries;
  std::vector<int32_t> band_slots;
  int mel_iters = 0, n_slots = 0;
};
bool fast512_applicable(int Nfft, int N);
int fast512_tile_frames();
// 0 on success, -1 if this configuration cannot use the fast kernel
int fast512_build_host(const MelBank &mel, int n_mfcc, double win_offset, int pad_left, int H, Fast512Host &h);
hipError_t launch_mfcc512(const LldParams &P, const Fast512Tables &F, hipStream_t s);
hipError_t launch_mfcc_generic(const LldParams &P, hipStream_t s);
hipError_t launch_delta(const DeltaParams &P, hipStream_t s);
}  // namespace smilehip
