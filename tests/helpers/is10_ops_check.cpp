// Test helper (tests/test_is10_ops_host.py): opensmile_amd/csrc/lld_is10_ops.hpp -- the per-frame bodies the kernels of
// lld_stage4_kernels.hip run one thread per frame -- compiled for the host, so that their arithmetic can be held against the real
// binary's levels without a GPU.
#include <cstdint>

#include "../../opensmile_amd/csrc/lld_is10_ops.hpp"

using namespace smilehip::is10;

extern "C" void is10_intensity_rows(const float *src, int64_t ld, int64_t n, int n_sum, const double *win, double win_sum, int flags,
                                    float *dst, int64_t ldd) {
  for (int64_t i = 0; i < n; ++i) intensity_frame(src + i * ld, n_sum, win, win_sum, flags, dst + i * ldd);
}
extern "C" void is10_lsp_rows(const float *lpc, int64_t ld, int64_t n, int p, float *dst, int64_t ldd) {
  for (int64_t i = 0; i < n; ++i) lsp_frame(lpc + i * ld, p, dst + i * ldd);
}
// returns the rows written (dst rows are consecutive)
extern "C" int64_t is10_pitch_smoother_rows(const float *src, int64_t ld, int64_t n, int n_cand, float cutoff, int oct, int simple, int flags,
                                            float *dst, int64_t ldd) {
  PitchSmootherOpts o{n_cand, oct, simple, flags, cutoff};
  PitchSmootherState s;
  pitch_smoother_reset(s);
  int64_t w = 0;
  for (int64_t i = 0; i < n; ++i)
    if (pitch_smoother_frame(o, s, src + i * ld, 1, dst + w * ldd) > 0) ++w;
  return w;
}
extern "C" void is10_vecop(int op, float aux, float logfloor, const float *src, float *dst, int64_t n) {
  for (int64_t i = 0; i < n; ++i) dst[i] = vecop(op, aux, logfloor, src[i]);
}
extern "C" void is10_vecop_reduce(int op, const float *src, int64_t ld, int64_t n_cols, int64_t n, float *dst) {
  for (int64_t i = 0; i < n; ++i) dst[i] = vecop_reduce(op, src + i * ld, n_cols);
}
