// cPitchACF's causal F0 contour (R10, src/lldcore/pitchACF.cpp:189-243) as one device function over an explicit state:
// the batch chain runs it over whole utterances (lld_pitch_smooth, one thread per utterance), the plugin's cPitchACF
// override frame by frame (lld_pitch_contour_step, smilehip_pitchacf_contour_step) -- one implementation for both.
//
// What the contour does to the raw pitch p of a frame (0 = unvoiced), with `last` / `before_last` the tracked pitches
// of the two frames before, `mean` a running mean of the voiced raw pitches and `edge` the kind of the latest
// voiced / unvoiced change (+1 onset, -1 offset, 0 none):
//   * a one-frame blip is dropped: an onset directly followed by an unvoiced frame clears `last`;
//   * a voiced value outside mean * (1 -+ 0.4) is replaced by the mean (and pulls the mean at a third of the usual 0.3);
//   * at an edge a `last` above the new value decays by 0.85; directly after an offset `last` takes the new value;
//   * the emitted F0 is the mean of the two previous tracked values (the previous one alone while either is zero), i.e.
//     the contour runs one to two frames behind.
// All arithmetic in float, in the reference's order: the outputs are the binary's bits.
#pragma once
#include <hip/hip_runtime.h>

namespace smilehip {

struct PitchContour {
  float last, before_last, mean, env;
  int edge;
  int pad[3];
};
static_assert(sizeof(PitchContour) == 32, "smilehip_pitchacf_contour_step's d_state is 8 words");

// returns the contour value the reference emits for this frame; S.env follows it (F0env, :236-240)
__device__ __forceinline__ float pitch_contour_step(PitchContour &S, float p) {
  const bool was = S.last > 0.0f, now = p > 0.0f;
  if (was == now) S.edge = 0;
  else if (now) S.edge = 1;
  else if (S.edge == 0) S.edge = -1;
  if (!now && S.edge == 1) S.last = 0.0f;
  float tracked = p;
  float pull = 0.3f;
  if (now) {
    const float spread = 0.4f;
    if (S.mean == 0.0f) S.mean = p;
    const bool inside = (p < (1.0f + spread) * S.mean) && (p > (1.0f - spread) * S.mean);
    if (!inside) { tracked = S.mean; pull /= 3.0f; }
    if (S.edge != 0 && S.last > tracked) S.last *= 0.85f;
    if (S.edge == -1) S.last = tracked;
    S.mean = (1.0f - pull) * S.mean + pull * p;
  }
  const float out = (S.before_last != 0.0f && S.last != 0.0f) ? 0.5f * (S.before_last + S.last) : S.last;
  S.before_last = S.last;
  S.last = tracked;
  if (out > 0.0f) S.env = 0.75f * S.env + 0.25f * out;
  return out;
}

}  // namespace smilehip
