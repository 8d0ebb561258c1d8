// libsmilehip, C ABI part 4: the per-component batched operators (what the plugin's overrides call).
#include "smilehip_internal.hpp"

#include <array>
#include <map>
#include <mutex>

// ---------------------------------------------- per-component entry points
#include "lld_stage.hpp"

#define STAGE_RET(expr, what)                                                              \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(SMILEHIP_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e_)); \
    return SMILEHIP_OK;                                                                    \
  } while (0)

static int check_frames(const void *s, const void *d, int64_t lds, int64_t ldd, int64_t nF, int64_t ws, int64_t wd,
                        const char *fn) {
  if (nF < 0 || (nF > 0 && (!s || !d))) return fail(SMILEHIP_ERR_INVALID, "%s: null pointer", fn);
  if (lds < ws || ldd < wd) return fail(SMILEHIP_ERR_INVALID, "%s: leading dimension too small", fn);
  return SMILEHIP_OK;
}

extern "C" int smilehip_pcm16_to_float(smilehip_context *ctx, const int16_t *d_pcm, int64_t n, float *d_out, void *stream) {
  if (!ctx || n < 0 || (n > 0 && (!d_pcm || !d_out))) return fail(SMILEHIP_ERR_INVALID, "smilehip_pcm16_to_float: bad argument");
  STAGE_RET(stage_pcm16(d_pcm, n, d_out, (hipStream_t)stream), "pcm16_to_float");
}

extern "C" int smilehip_pcm_convert(smilehip_context *ctx, const void *d_raw, int n_bps, int n_bits, int n_chan,
                                    int mono_mixdown, int64_t n, float *d_out, void *stream) {
  if (!ctx || n < 0 || n_chan < 1 || (n > 0 && (!d_raw || !d_out)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pcm_convert: bad argument");
  if (n_bps < 1 || n_bps > 4 || (n_bps == 4 && n_bits != 24 && n_bits != 32))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pcm_convert: unknown sample format (nBPS=%d, nBits=%d)", n_bps, n_bits);
  STAGE_RET(stage_pcm_convert(d_raw, n_bps, n_bits, n_chan, mono_mixdown != 0, n, d_out, (hipStream_t)stream), "pcm_convert");
}

extern "C" int smilehip_pcm_convert_float(smilehip_context *ctx, const float *d_raw, int n_chan, int mono_mixdown, int64_t n,
                                          float *d_out, void *stream) {
  if (!ctx || n < 0 || n_chan < 1 || (n > 0 && (!d_raw || !d_out)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pcm_convert_float: bad argument");
  STAGE_RET(stage_pcm_convert_float(d_raw, n_chan, mono_mixdown != 0, n, d_out, (hipStream_t)stream), "pcm_convert_float");
}

// ---- R11, general option set: an operator object (the band edges, the slope's sums and the sharpness weights are functions of the
// options and the frequency axis alone)
struct smilehip_spectral_op {
  smilehip_context *ctx = nullptr;
  SpectralGeneral G;
  DevBuf<double> d_sharp;
  int n_out = 0;
};
extern "C" int smilehip_spectral_op_destroy(smilehip_spectral_op *op) {
  delete op;
  return SMILEHIP_OK;
}
extern "C" int smilehip_spectral_op_n_out(const smilehip_spectral_op *op) { return op ? op->n_out : -1; }
extern "C" int smilehip_spectral_opts_count(const smilehip_spectral_opts *o) {
  if (!o || o->n_bands < 0 || o->n_bands > 16 || o->n_rolloff < 0 || o->n_rolloff > 16 || o->n_slopes < 0 || o->n_slopes > 16) return -1;
  return o->n_bands + o->n_slopes + o->n_rolloff + (o->spec_diff != 0) + (o->spec_pos_diff != 0) + (o->flux != 0) + (o->flux_centroid != 0) +
         (o->flux_at_flux_centroid != 0) + (o->centroid != 0) + (o->max_pos != 0) + (o->min_pos != 0) + (o->entropy != 0) +
         (o->standard_deviation != 0) + (o->variance != 0) + (o->skewness != 0) + (o->kurtosis != 0) + (o->slope != 0) +
         (o->sharpness != 0) + (o->harmonicity != 0) + (o->flatness != 0);
}
extern "C" int smilehip_spectral_op_create(smilehip_context *ctx, const smilehip_spectral_opts *o, int64_t K, double frame_size_sec,
                                           smilehip_spectral_op **out) {
  if (!ctx || !o || !out) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_op_create: null argument");
  const int n_out = smilehip_spectral_opts_count(o);
  if (n_out < 1) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_op_create: 0 .. 16 bands, 0 .. 16 rollOff points, at least one output");
  if (K < 8 || K > (1 << 20) || !(frame_size_sec > 0.0)) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_op_create: K %lld, frame size %g s", (long long)K, frame_size_sec);
  for (int b = 0; b < o->n_bands; ++b)
    if (o->band_lo[b] < 0 || o->band_hi[b] <= o->band_lo[b]) return fail(SMILEHIP_ERR_INVALID, "cSpectral bands[%d] = %d-%d", b, o->band_lo[b], o->band_hi[b]);
  for (int i = 0; i < o->n_rolloff; ++i)
    if (!(o->rolloff[i] >= 0.0 && o->rolloff[i] <= 1.0)) return fail(SMILEHIP_ERR_INVALID, "cSpectral rollOff[%d] = %g", i, o->rolloff[i]);
  smilehip_spectral_op *op = new smilehip_spectral_op;
  op->ctx = ctx;
  op->n_out = n_out;
  SpectralGeneral &G = op->G;
  std::memset(&G, 0, sizeof(G));
  G.K = (int32_t)K;
  G.frame_size_sec = frame_size_sec;
  G.n_bands = o->n_bands; G.n_rolloff = o->n_rolloff;
  for (int i = 0; i < o->n_rolloff; ++i) G.rolloff[i] = o->rolloff[i];
  G.flux = o->flux != 0; G.centroid = o->centroid != 0; G.max_pos = o->max_pos != 0; G.min_pos = o->min_pos != 0;
  G.entropy = o->entropy != 0; G.variance = o->variance != 0; G.skewness = o->skewness != 0; G.kurtosis = o->kurtosis != 0;
  G.slope = o->slope != 0; G.sharpness = o->sharpness != 0; G.harmonicity = o->harmonicity != 0;
  G.flatness = o->flatness != 0; G.log_flatness = o->log_flatness != 0;
  G.spec_diff = o->spec_diff != 0; G.spec_pos_diff = o->spec_pos_diff != 0; G.flux_centroid = o->flux_centroid != 0;
  G.flux_at_flux_centroid = o->flux_at_flux_centroid != 0; G.standard_deviation = o->standard_deviation != 0;
  G.n_out = n_out; G.n_slopes = o->n_slopes;
  const int Nsrc = (int)K;
  const double F0 = 1.0 / frame_size_sec;                // frq[i] = F0 * i, transformFft.cpp:102-117
  for (int b = 0; b < o->n_bands + o->n_slopes; ++b) {    // a band's (spectral.cpp:779-826) or a slope band's (:872-943: the same mapping) edge bins and weights
    const bool is_slope = b >= o->n_bands;
    const int lo = is_slope ? o->slope_lo[b - o->n_bands] : o->band_lo[b], hi = is_slope ? o->slope_hi[b - o->n_bands] : o->band_hi[b];
    if (is_slope && (lo < 0 || hi <= lo)) { delete op; return fail(SMILEHIP_ERR_INVALID, "cSpectral slopes[%d] = %d-%d", b - o->n_bands, lo, hi); }
    int ii;
    double wghtL, wghtR, idxL, idxR;
    for (ii = 0; ii < Nsrc; ii++) if (F0 * ii > (double)lo) break;
    if ((ii < Nsrc) && (ii > 0)) wghtL = (F0 * ii - (double)lo) / (F0 * ii - F0 * (ii - 1)); else wghtL = 1.0;
    idxL = (double)ii - 1.0;
    if (idxL < 0) idxL = 0;
    if (idxL >= Nsrc) idxL = Nsrc;
    if (wghtL == 0.0) wghtL = 1.0;
    for (ii = 0; ii < Nsrc; ii++) if (F0 * ii >= (float)hi) break;
    if ((ii < Nsrc) && (ii > 0)) wghtR = ((double)hi - F0 * (ii - 1)) / (F0 * ii - F0 * (ii - 1)); else wghtR = 1.0;
    if ((ii < Nsrc) && (F0 * ii == (float)hi)) idxR = (double)ii; else idxR = (double)ii - 1.0;
    if (idxR >= Nsrc) idxR = Nsrc - 1;
    if (wghtR == 0.0) wghtR = 1.0;
    int iL = (int)std::floor(idxL), iR = (int)std::floor(idxR);
    if (iL >= Nsrc) { iL = iR = Nsrc - 1; wghtR = 0.0; wghtL = 0.0; }
    if (iR >= Nsrc) { iR = Nsrc - 1; wghtR = 1.0; }
    if (iL < 0) iL = 0;
    if (iR < 0) iR = 0;
    if (iR < iL) { delete op; return fail(SMILEHIP_ERR_INVALID, "cSpectral %s[%d] = %d-%d lies between two bins of this spectrum", is_slope ? "slopes" : "bands", is_slope ? b - o->n_bands : b, lo, hi); }
    if (is_slope) {
      const int k = b - o->n_bands;
      G.sl_iL[k] = iL; G.sl_iR[k] = iR; G.sl_wL[k] = wghtL; G.sl_wR[k] = wghtR; G.sl_Nind[k] = idxR - idxL;
    } else {
      G.band_iL[b] = iL; G.band_iR[b] = iR; G.band_wL[b] = wghtL; G.band_wR[b] = wghtR;
    }
  }
  for (int64_t i = 1; i < K; ++i) { G.slope_S2f += (F0 * i) * (F0 * i); G.slope_Sf += F0 * i; }
  std::vector<double> sw((size_t)(K - 1));                // sharpness weights bark(f) g(bark(f)), spectral.cpp:1440-1455, smileUtil.c:1063-1078, 1123-1137
  for (int64_t j = 1; j < K; ++j) {
    const double x = F0 * double(j);
    double zz = 0.0;
    if (x > 0) {
      zz = (26.81 / (1.0 + 1960.0 / x)) - 0.53;
      if (zz < 2) zz = 0.85 * zz + 0.3;
      else if (zz > 20.1) zz = 1.22 * zz - 0.22 * 20.1;
    }
    const double g = (zz <= 16.0) ? 1.0 : std::pow((zz - 16.0) / 4.0, 1.5849625) + 1.0;
    sw[(size_t)(j - 1)] = zz * g;
  }
  const int rc = op->d_sharp.upload(sw);
  if (rc) { delete op; return rc; }
  G.sharp_w = op->d_sharp.p;
  *out = op;
  return SMILEHIP_OK;
}
extern "C" int smilehip_spectral_op_frames(smilehip_spectral_op *op, const float *d_mag, int64_t ld_src, float *d_state, int first,
                                           float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  if (!op) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_op_frames: null operator");
  if (n_frames < 0 || ld_src < op->G.K || ld_dst < op->n_out || (n_frames > 0 && (!d_mag || !d_dst)) ||
      ((op->G.flux || op->G.spec_diff || op->G.spec_pos_diff || op->G.flux_centroid || op->G.flux_at_flux_centroid) && !d_state))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_op_frames: bad argument (K %d, %d outputs; d_state is needed when flux is on)", op->G.K, op->n_out);
  STAGE_RET(stage_spectral_general(op->G, d_mag, ld_src, d_state, first, d_dst, ld_dst, n_frames, (hipStream_t)stream), "spectral (general)");
}

extern "C" int smilehip_preemphasis_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, float *d_dst,
                                           int64_t ld_dst, int64_t n_frames, int64_t N, float k, int de, void *stream) {
  if (!ctx || N < 1) return fail(SMILEHIP_ERR_INVALID, "smilehip_preemphasis_frames: bad argument");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, N, N, "smilehip_preemphasis_frames");
  if (rc) return rc;
  STAGE_RET(stage_preemph(d_src, ld_src, d_dst, ld_dst, n_frames, N, k, de, (hipStream_t)stream), "preemphasis");
}

extern "C" int smilehip_window_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                      int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_window_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (!(p->stage_mask & SMILEHIP_STAGE_WINDOW)) return fail(SMILEHIP_ERR_INVALID, "smilehip_window_frames: plan was built without this stage's tables");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.N, p->geo.N, "smilehip_window_frames");
  if (rc) return rc;
  STAGE_RET(stage_window(d_src, ld_src, d_dst, ld_dst, n_frames, p->geo.N, p->d_window.p, (float)p->cfg.win_offset,
                         (hipStream_t)stream), "window");
}

extern "C" int smilehip_rfft_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                    int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_rfft_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.N, p->geo.Nfft, "smilehip_rfft_frames");
  if (rc) return rc;
  const int pad = p->cfg.zero_pad_symmetric ? (int)((p->geo.Nfft - p->geo.N) / 2) : 0;
  if (!p->fft_radix2)
    STAGE_RET(stage_rfft_oo(d_src, ld_src, d_dst, ld_dst, n_frames, (int)p->geo.N, (int)p->geo.Nfft, pad, p->oo.tab(),
                            (hipStream_t)stream), "rfft");
  STAGE_RET(stage_rfft(d_src, ld_src, d_dst, ld_dst, n_frames, (int)p->geo.N, (int)p->geo.Nfft, pad, p->d_tw_half.p,
                       p->d_tw_full.p, (hipStream_t)stream), "rfft");
}

extern "C" int smilehip_sumsq_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int64_t n_frames,
                                     double *d_out, void *stream) {
  if (!ctx || N < 1 || n_frames < 0 || ld_src < N || (n_frames > 0 && (!d_src || !d_out)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_sumsq_frames: bad argument");
  STAGE_RET(stage_sumsq(d_src, ld_src, N, n_frames, d_out, (hipStream_t)stream), "sumsq");
}

extern "C" int smilehip_zcr_count_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N,
                                         int64_t n_frames, int32_t *d_out, void *stream) {
  if (!ctx || N < 1 || n_frames < 0 || ld_src < N || (n_frames > 0 && (!d_src || !d_out)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_zcr_count_frames: bad argument");
  STAGE_RET(stage_zcr_count(d_src, ld_src, N, n_frames, d_out, (hipStream_t)stream), "zcr_count");
}

extern "C" int smilehip_valbased_select_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int64_t n_frames,
                                              int32_t idx, float threshold, int32_t invert, int32_t allow_equal, int32_t zero_vec,
                                              int32_t remove_idx, float output_val, float *d_dst, int64_t ld_dst, int32_t *d_keep,
                                              void *stream) {
  const int64_t n_out = remove_idx ? N - 1 : N;
  if (!ctx || N < 1 || N > (1 << 20) || n_out < 1 || idx < 0 || n_frames < 0 || ld_src < N || ld_dst < n_out ||
      (n_frames > 0 && (!d_src || !d_dst || !d_keep)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_valbased_select_frames: bad argument");
  STAGE_RET(stage_valbased(d_src, ld_src, (int)N, n_frames, idx, threshold, invert, allow_equal, zero_vec, remove_idx, output_val, d_dst,
                           ld_dst, d_keep, (hipStream_t)stream), "valbased_select");
}

extern "C" int smilehip_acf_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                   int64_t n_out, int64_t n_frames, int use_power, int cepstrum, int norm_output,
                                   int abs_cepstrum, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_acf_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (n_out < 1 || n_out > p->geo.Nfft / 2)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_acf_frames: n_out %lld outside [1, %lld] (symmetric half of the inverse FFT)",
                (long long)n_out, (long long)(p->geo.Nfft / 2));
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.K, n_out, "smilehip_acf_frames");
  if (rc) return rc;
  STAGE_RET(stage_acf(d_src, ld_src, d_dst, ld_dst, n_frames, (int)p->geo.K, (int)n_out, use_power, cepstrum, norm_output,
                      abs_cepstrum, p->d_tw_half.p, p->d_tw_full.p, p->fft_radix2 ? OouraTab{} : p->oo.tab(), (hipStream_t)stream), "acf");
}

extern "C" int smilehip_pitchacf_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t n, int64_t n_frames,
                                        double fs_sec, double max_pitch, double *d_voicing, int32_t *d_max_idx, void *stream) {
  if (!ctx || n < 4 || n_frames < 0 || ld_src < 2 * n || (n_frames > 0 && (!d_src || !d_voicing || !d_max_idx)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pitchacf_frames: bad argument");
  STAGE_RET(stage_pitchacf(d_src, ld_src, n_frames, (int)n, fs_sec, max_pitch, d_voicing, d_max_idx, (hipStream_t)stream), "pitchacf");
}

extern "C" int smilehip_pitchacf_contour_step(smilehip_context *ctx, const double *d_voicing, const int32_t *d_max_idx, double t_samp,
                                              double voicing_cutoff, float *d_state, float *d_out4, void *stream) {
  if (!ctx || !d_voicing || !d_max_idx || !d_state || !d_out4 || !(t_samp > 0.0))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pitchacf_contour_step: bad argument");
  STAGE_RET(launch_pitch_contour_step(d_voicing, d_max_idx, t_samp, voicing_cutoff, d_state, d_out4, (hipStream_t)stream), "pitchacf contour");
}

extern "C" int smilehip_pitchacf_contour_frames(smilehip_context *ctx, const double *d_voicing, const int32_t *d_max_idx, double t_samp,
                                                double voicing_cutoff, float *d_state, float *d_out4, int64_t n_frames, void *stream) {
  if (!ctx || n_frames < 0 || !d_voicing || !d_max_idx || !d_state || !d_out4 || !(t_samp > 0.0))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pitchacf_contour_frames: bad argument");
  STAGE_RET(launch_pitch_contour_frames(d_voicing, d_max_idx, t_samp, voicing_cutoff, d_state, d_out4, n_frames, (hipStream_t)stream), "pitchacf contour");
}

extern "C" int smilehip_spectral_frames(smilehip_plan *p, const float *d_mag, int64_t ld_src, float *d_state, int first,
                                        float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (!p->d_sharp.p) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: plan was built without SMILEHIP_STAGE_SPECTRAL");
  if (p->geo.K != 129 && p->geo.K != 257 && p->geo.K != 513)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: K = %lld, the kernels cover K = 129 / 257 / 513 (20 ms frames at 8 .. 48 kHz)", (long long)p->geo.K);
  if (!d_state && n_frames > 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: null state buffer");
  int rc = check_frames(d_mag, d_dst, ld_src, ld_dst, n_frames, p->geo.K, 15, "smilehip_spectral_frames");
  if (rc) return rc;
  SpectralConsts C;
  C.log_tab = nullptr;
  C.fsSec = p->geo.fft_frame_size_sec;
  C.sharp_w = p->d_sharp.p;
  for (int i = 0; i < 2; ++i) {
    C.band_iL[i] = p->band_iL[i]; C.band_iR[i] = p->band_iR[i];
    C.band_wL[i] = p->band_wL[i]; C.band_wR[i] = p->band_wR[i];
  }
  C.slope_Sf = p->slope_Sf;
  C.slope_S2f = p->slope_S2f;
  STAGE_RET(stage_spectral(d_mag, ld_src, d_state, first != 0, d_dst, ld_dst, n_frames, (int)p->geo.K, C, (hipStream_t)stream), "spectral");
}

extern "C" int smilehip_plp_audspec_frames(smilehip_context *ctx, const float *d_mel, int64_t ld_src, int n_bands,
                                           const float *d_eql, float melfloor, float compression, int new_rasta,
                                           const float *rasta_coef, float *d_state, float *d_dst, int64_t ld_dst,
                                           int64_t n_frames, void *stream) {
  if (!ctx || n_bands < 1 || n_bands > 64 || !d_eql) return fail(SMILEHIP_ERR_INVALID, "smilehip_plp_audspec_frames: bad argument (1..64 bands)");
  if (new_rasta && (!rasta_coef || !d_state)) return fail(SMILEHIP_ERR_INVALID, "smilehip_plp_audspec_frames: RASTA needs coefficients and a state buffer");
  int rc = check_frames(d_mel, d_dst, ld_src, ld_dst, n_frames, n_bands, n_bands, "smilehip_plp_audspec_frames");
  if (rc) return rc;
  PlpConsts Q;
  Q.melfloor = melfloor;
  Q.compression = compression;
  Q.iir = new_rasta ? rasta_coef[0] : 0.0f;
  for (int i = 0; i < 5; ++i) Q.fir[i] = new_rasta ? rasta_coef[1 + i] : 0.0f;
  if (new_rasta < 0 || new_rasta > 2) return fail(SMILEHIP_ERR_INVALID, "smilehip_plp_audspec_frames: rasta mode must be 0 (none), 1 (newRASTA) or 2 (RASTA)");
  STAGE_RET(stage_plp(d_mel, ld_src, n_bands, d_eql, Q, new_rasta, d_state, d_dst, ld_dst, n_frames, (hipStream_t)stream), "plp");
}

extern "C" int smilehip_plp_cc_frames(smilehip_context *ctx, const float *d_mel, int64_t ld_src, int n_bands, const float *d_eql,
                                      float melfloor, float compression, int lp_order, const float *d_cos, const float *d_sin,
                                      float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  if (!ctx || n_bands < 2 || n_bands > 64 || lp_order < 1 || lp_order > 15 || !d_eql || !d_cos || !d_sin)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_plp_cc_frames: bad argument (2..64 bands, lpOrder 1..15)");
  int rc = check_frames(d_mel, d_dst, ld_src, ld_dst, n_frames, n_bands, lp_order + 1, "smilehip_plp_cc_frames");
  if (rc) return rc;
  STAGE_RET(stage_plp_cc(d_mel, ld_src, n_bands, d_eql, melfloor, compression, lp_order, d_cos, d_sin, d_dst, ld_dst, n_frames,
                         (hipStream_t)stream), "plp_cc");
}

extern "C" int smilehip_plp_stage_frames(smilehip_context *ctx, const float *d_mel, int64_t ld_src, int n_bands, const float *d_eql,
                                         float melfloor, float compression, int lp_order, const float *d_cos, int out_stage,
                                         float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  if (!ctx || n_bands < 2 || n_bands > 64 || lp_order < 1 || lp_order > 15 || !d_eql || !d_cos || (out_stage != 1 && out_stage != 2))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_plp_stage_frames: bad argument (2..64 bands, lpOrder 1..15, stage 1 | 2)");
  int rc = check_frames(d_mel, d_dst, ld_src, ld_dst, n_frames, n_bands, out_stage == 1 ? lp_order + 1 : lp_order, "smilehip_plp_stage_frames");
  if (rc) return rc;
  STAGE_RET(stage_plp_cc(d_mel, ld_src, n_bands, d_eql, melfloor, compression, lp_order, d_cos, d_cos /* no lifter here */, d_dst, ld_dst,
                         n_frames, (hipStream_t)stream, out_stage), "plp_stage");
}

extern "C" int smilehip_window_op_row(smilehip_context *ctx, const float *d_x, float *d_y, int64_t n_t, int kind, int W,
                                      void *stream) {
  if (!ctx || n_t < 0 || W < 1 || (kind != 0 && kind != 1) || (n_t > 0 && (!d_x || !d_y)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_window_op_row: bad argument");
  STAGE_RET(stage_window_op(d_x, d_y, n_t, kind, W, delta_norm(W), (hipStream_t)stream), "window_op");
}

extern "C" int smilehip_window_op_row_ex(smilehip_context *ctx, const float *d_x, float *d_y, int64_t n_t, int kind, int W,
                                         float *d_norm_io, void *stream) {
  if (kind == 0 || kind == 1) return smilehip_window_op_row(ctx, d_x, d_y, n_t, kind, W, stream);
  if (!ctx || n_t < 0 || W < 1 || (kind != 2 && kind != 3) || (kind == 3 && !d_norm_io) || (n_t > 0 && (!d_x || !d_y)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_window_op_row_ex: bad argument");
  STAGE_RET(stage_window_op_seq(d_x, d_y, n_t, kind, W, d_norm_io, (hipStream_t)stream), "window_op_seq");
}

extern "C" int smilehip_delta_op_row(smilehip_context *ctx, const float *d_x, float *d_y, int64_t n_t, int W, int flags, float *d_norm_io,
                                     void *stream) {
  if (!ctx || n_t < 0 || W < 0 || W > 64 || (flags & ~15) || ((flags & 8) && W > 0 && !d_norm_io) || (n_t > 0 && (!d_x || !d_y)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_op_row: bad argument (deltawin 0 .. 64, flags 1 | 2 | 4 | 8, d_norm_io with onlyInSegments)");
  STAGE_RET(stage_delta_op(d_x, d_y, n_t, W, W > 0 ? delta_norm(W) : 1.0f, flags, d_norm_io, (hipStream_t)stream), "delta_op");
}

extern "C" int smilehip_window_op_block(smilehip_context *ctx, const float *d_x, int64_t ld_x, float *d_y, int64_t ld_y, int64_t n_t,
                                        int32_t n_cols, int op, int W, int delta_flags, void *stream) {
  if (!ctx || n_t < 0 || n_cols < 1 || ld_x < n_cols || ld_y < n_cols || op < 0 || op > 2 || W < (op == 0 ? 0 : 1) || W > 64 ||
      (delta_flags & ~7) || (n_t > 0 && (!d_x || !d_y)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_window_op_block: bad argument (op 0 | 1 | 2, W <= 64, delta flags 1 | 2 | 4)");
  STAGE_RET(stage_window_op_block(d_x, ld_x, d_y, ld_y, n_t, n_cols, op, W, (op == 0 && W > 0) ? delta_norm(W) : 1.0f,
                                  op == 0 ? delta_flags : 0, (hipStream_t)stream), "window_op_block");
}

extern "C" int smilehip_delta_segments_block(smilehip_context *ctx, const float *d_x, int64_t ld_x, float *d_y, int64_t ld_y, int64_t n_ticks,
                                             int32_t tick_frames, int32_t n_cols, int W, int delta_flags, float *d_norm_io, void *stream) {
  if (!ctx || n_ticks < 0 || tick_frames < 1 || n_cols < 1 || ld_x < n_cols || ld_y < n_cols || W < 0 || W > 64 || (delta_flags & ~7) || !d_norm_io ||
      (n_ticks > 0 && (!d_x || !d_y)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_segments_block: bad argument");
  STAGE_RET(stage_delta_seg_block(d_x, ld_x, d_y, ld_y, n_ticks, tick_frames, n_cols, W, d_norm_io, delta_flags | 8, (hipStream_t)stream), "delta_seg_block");
}

extern "C" int smilehip_frame_rows(smilehip_context *ctx, const float *d_samples, int64_t frame_size, int64_t frame_step, int64_t n_frames,
                                   float *d_dst, int64_t ld_dst, void *stream) {
  if (!ctx || frame_size < 1 || frame_step < 1 || n_frames < 0 || ld_dst < frame_size || (n_frames > 0 && (!d_samples || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_frame_rows: bad argument");
  STAGE_RET(stage_frame_rows(d_samples, frame_size, frame_step, n_frames, d_dst, ld_dst, (hipStream_t)stream), "frame_rows");
}

extern "C" int smilehip_fftmag_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                      int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_fftmag_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.Nfft, p->geo.K, "smilehip_fftmag_frames");
  if (rc) return rc;
  STAGE_RET(stage_fftmag(d_src, ld_src, d_dst, ld_dst, n_frames, (int)p->geo.Nfft, (hipStream_t)stream), "fftmag");
}

extern "C" int smilehip_irfft_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                     int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_irfft_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.Nfft, p->geo.Nfft, "smilehip_irfft_frames");
  if (rc) return rc;
  if (!p->oo.d_tw.p) return fail(SMILEHIP_ERR_INVALID, "smilehip_irfft_frames: the reference-order transform is built for 64 .. 8192 points");
  STAGE_RET(stage_irfft_oo(d_src, ld_src, d_dst, ld_dst, n_frames, (int)p->geo.Nfft, p->oo.tab(), (hipStream_t)stream), "irfft");
}

extern "C" int smilehip_fftmagphase_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t nfft, int32_t flags,
                                           float dbp_norm, float min_dbp, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  const int64_t K = nfft / 2 + 1;
  const int64_t n_out = ((flags & 1) ? K : 0) + ((flags & 2) ? K : 0);
  if (!ctx || nfft < 4 || (nfft & 1) || !(flags & 3) || (flags & ~31) || n_frames < 0 || ld_src < nfft || ld_dst < n_out ||
      (n_frames > 0 && (!d_src || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_fftmagphase_frames: bad argument (flags: 1 magnitude, 2 phase, 4 normalise, 8 power, 16 dBpsd)");
  STAGE_RET(stage_fftmagphase(d_src, ld_src, d_dst, ld_dst, n_frames, (int)nfft, flags, dbp_norm, min_dbp, (hipStream_t)stream), "fftmagphase");
}

extern "C" int smilehip_melspec_table_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t K, int32_t n_bands, int32_t dense,
                                             const float *d_coef, const int32_t *d_chanmap, int32_t n_lo, int32_t n_hi, int32_t use_power,
                                             float htk_scale, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  if (!ctx || K < 2 || K > 8193 || n_bands < 1 || n_bands > 4096 || !d_coef || !d_chanmap || n_lo < 0 || n_hi > K || n_frames < 0 ||
      ld_src < K || ld_dst < n_bands || (n_frames > 0 && (!d_src || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_melspec_table_frames: bad argument");
  STAGE_RET(stage_melspec_table(d_src, ld_src, (int)K, n_bands, dense ? 1 : 0, d_coef, d_chanmap, n_lo, n_hi, use_power, htk_scale, d_dst, ld_dst,
                                n_frames, (hipStream_t)stream), "melspec_table");
}

extern "C" int smilehip_melspec_inverse_table_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int32_t n_src, int64_t K,
                                                     const float *d_coef, const int32_t *d_chanmap, int32_t n_lo, int32_t n_hi,
                                                     int32_t use_power, float htk_div, float *d_dst, int64_t ld_dst, int64_t n_frames,
                                                     void *stream) {
  if (!ctx || K < 2 || K > 8193 || n_src < 1 || n_src > 4096 || !d_coef || !d_chanmap || n_lo < 0 || n_hi < 0 || n_frames < 0 ||
      ld_src < n_src || ld_dst < K || !(htk_div > 0.0f) || (n_frames > 0 && (!d_src || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_melspec_inverse_table_frames: bad argument");
  STAGE_RET(stage_melspec_inverse_table(d_src, ld_src, n_src, (int)K, d_coef, d_chanmap, n_lo, n_hi, use_power, htk_div, d_dst, ld_dst,
                                        n_frames, (hipStream_t)stream), "melspec_inverse_table");
}

extern "C" int smilehip_pitchacf_zcr_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t n, int64_t n_frames,
                                            double fs_sec, double max_pitch, double *d_zcr, void *stream) {
  if (!ctx || n < 2 || n > (1 << 20) || n_frames < 0 || ld_src < n || !(fs_sec > 0.0) || (n_frames > 0 && (!d_src || !d_zcr)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pitchacf_zcr_frames: bad argument");
  // preskip of cPitchACF::processVector (pitchACF.cpp:151-158): Nd = 2 n values in [acf | cepstrum], Tsamp = fsSec / Nd
  const double Tsamp = fs_sec / (double)(2 * n);
  int skip = (max_pitch <= 0.0) ? 0 : (int)(1.0 / (max_pitch * Tsamp));
  if (skip < 0 || skip >= n) return fail(SMILEHIP_ERR_INVALID, "smilehip_pitchacf_zcr_frames: maxPitch leaves no lag to search (preskip %d of %lld)", skip, (long long)n);
  STAGE_RET(stage_pitchacf_zcr(d_src, ld_src, n_frames, (int)n, skip, d_zcr, (hipStream_t)stream), "pitchacf_zcr");
}

extern "C" int smilehip_mzcr_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int64_t n_frames, int32_t flags,
                                    float *d_dst, int64_t ld_dst, void *stream) {
  const int n_out = ((flags & 1) ? 1 : 0) + ((flags & 2) ? 1 : 0) + ((flags & 4) ? 1 : 0) + ((flags & 8) ? 2 : 0) + ((flags & 16) ? 1 : 0);
  if (!ctx || N < 1 || N > (1 << 15) || !(flags & 31) || (flags & ~31) || n_frames < 0 || ld_src < N || ld_dst < n_out ||
      (n_frames > 0 && (!d_src || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_mzcr_frames: bad argument (flags: 1 zcr, 2 mcr, 4 amax, 8 maxmin, 16 dc; N <= 32768)");
  STAGE_RET(stage_mzcr(d_src, ld_src, (int)N, n_frames, flags, d_dst, ld_dst, (hipStream_t)stream), "mzcr");
}

extern "C" int smilehip_melspec_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                       int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_melspec_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (!(p->stage_mask & SMILEHIP_STAGE_MEL)) return fail(SMILEHIP_ERR_INVALID, "smilehip_melspec_frames: plan was built without this stage's tables");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.K, p->mel.n_bands, "smilehip_melspec_frames");
  if (rc) return rc;
  STAGE_RET(stage_melspec(d_src, ld_src, d_dst, ld_dst, n_frames, (int)p->geo.K, p->mel.n_bands, p->cfg.use_power,
                          p->d_mel_coef.p, p->d_mel_rng.p, p->mel.scale, (hipStream_t)stream), "melspec");
}

extern "C" int smilehip_mfcc_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                    int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (!(p->stage_mask & SMILEHIP_STAGE_MFCC)) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_frames: plan was built without this stage's tables");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->mel.n_bands, p->dct.n_mfcc, "smilehip_mfcc_frames");
  if (rc) return rc;
  STAGE_RET(stage_mfcc(d_src, ld_src, d_dst, ld_dst, n_frames, p->mel.n_bands, p->dct.n_mfcc, p->d_dct_rows.p,
                       p->d_dct_gain.p, p->dct.melfloor, p->dct.log_floor, (hipStream_t)stream), "mfcc");
}

extern "C" int smilehip_mfcc_inverse_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                            int64_t n_frames, int do_log, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_inverse_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (!(p->stage_mask & SMILEHIP_STAGE_MFCC)) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_inverse_frames: plan was built without this stage's tables");
  if (p->dct.n_mfcc < 1 || p->dct.n_mfcc > 64 || (int)p->dct.lifter.size() < p->dct.n_mfcc)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_inverse_frames: 1 .. 64 coefficients");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->dct.n_mfcc, p->mel.n_bands, "smilehip_mfcc_inverse_frames");
  if (rc) return rc;
  STAGE_RET(stage_mfcc_inverse(d_src, ld_src, d_dst, ld_dst, n_frames, p->mel.n_bands, p->dct.first, p->dct.last, p->cfg.mfcc_htk_compatible != 0,
                               do_log != 0, p->d_dct_rows.p, p->dct.lifter.data(), (hipStream_t)stream), "mfcc_inverse");
}

// ---------------------------------------------- F0 group, per-component
static int f0_rows(smilehip_plan *p, int mode, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst, int64_t n_frames,
                   void *stream, const char *fn) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "%s: null plan", fn);
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (p->cfg.chain_kind != SMILEHIP_CHAIN_COMPARE_F0) return fail(SMILEHIP_ERR_INVALID, "%s: plan is not an F0 chain plan", fn);
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.K, mode == 1 ? p->geo.K : 21, fn);
  if (rc || n_frames == 0) return rc;
  F0Params Q;
  fill_f0_params(p, Q);
  Q.mode = mode;
  Q.n_rows = n_frames;
  Q.in_rows = d_src;
  Q.ld_in = ld_src;
  if (mode == 1) { Q.hps_tap = d_dst; Q.ld_tap = ld_dst; }
  else { Q.shs = d_dst; Q.ld_shs = ld_dst; }
  STAGE_RET(launch_f0_rows(Q, p->ctx->prop.multiProcessorCount, (hipStream_t)stream), fn);
}

extern "C" int smilehip_specscale_frames(smilehip_plan *p, const float *d_mag, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                         int64_t n_frames, void *stream) {
  return f0_rows(p, 1, d_mag, ld_src, d_dst, ld_dst, n_frames, stream, "smilehip_specscale_frames");
}

extern "C" int smilehip_pitchshs_frames(smilehip_plan *p, const float *d_hps, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                        int64_t n_frames, void *stream) {
  return f0_rows(p, 2, d_hps, ld_src, d_dst, ld_dst, n_frames, stream, "smilehip_pitchshs_frames");
}

// ---------------------------------------------- cPitchSmootherViterbi as a stream (the plugin's tick-level override)
struct smilehip_viterbi_stream {
  smilehip_context *ctx = nullptr;
  F0Params Q;
  float *d_frames = nullptr;
  int64_t cap_frames = 0, n_frames = 0;
  int *d_st = nullptr, *d_paths = nullptr, *d_decided = nullptr;
  int64_t cap_decided = 128;                               // (frame, state) pairs d_decided holds
  double *d_dstate = nullptr;
};

extern "C" int smilehip_viterbi_stream_destroy(smilehip_viterbi_stream *s) {
  if (!s) return SMILEHIP_OK;
  if (s->d_frames) (void)hipFree(s->d_frames);
  if (s->d_st) (void)hipFree(s->d_st);
  if (s->d_paths) (void)hipFree(s->d_paths);
  if (s->d_decided) (void)hipFree(s->d_decided);
  if (s->d_dstate) (void)hipFree(s->d_dstate);
  delete s;
  return SMILEHIP_OK;
}

extern "C" int smilehip_viterbi_stream_create(smilehip_context *ctx, int32_t buffer_len, float voicing_cutoff, const double *weights6,
                                              smilehip_viterbi_stream **out) {
  if (!ctx || !weights6 || !out) return fail(SMILEHIP_ERR_INVALID, "smilehip_viterbi_stream_create: null argument");
  if (buffer_len < 2 || buffer_len > f0_viterbi_max_buffer())
    return fail(SMILEHIP_ERR_INVALID, "smilehip_viterbi_stream_create: bufferLength %d outside 2..%d", buffer_len, f0_viterbi_max_buffer());
  smilehip_viterbi_stream *s = new smilehip_viterbi_stream;
  s->ctx = ctx;
  std::memset(&s->Q, 0, sizeof(s->Q));
  s->Q.vit_buf = buffer_len;
  s->Q.n_cand = 6;
  s->Q.voicing_cutoff = voicing_cutoff;
  for (int i = 0; i < 6; ++i) s->Q.vit_w[i] = weights6[i];
  const int ns = f0_viterbi_states(), np = ns * f0_viterbi_max_buffer();
  std::vector<double> ds((size_t)ns + 1, 0.0);
  ds[(size_t)ns] = 1.0;                                    // lastChange starts at 1.0 (pitchSmootherViterbi.hpp)
  const int st0[4] = {0, -1, 0, 0};
  if (hipMalloc(reinterpret_cast<void **>(&s->d_st), sizeof(st0)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&s->d_paths), sizeof(int) * (size_t)np) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&s->d_decided), sizeof(int) * 2 * 128) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&s->d_dstate), sizeof(double) * ds.size()) != hipSuccess ||
      hipMemcpy(s->d_st, st0, sizeof(st0), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(s->d_paths, 0, sizeof(int) * (size_t)np) != hipSuccess ||
      hipMemcpy(s->d_dstate, ds.data(), sizeof(double) * ds.size(), hipMemcpyHostToDevice) != hipSuccess) {
    smilehip_viterbi_stream_destroy(s);
    return fail(SMILEHIP_ERR_HIP, "smilehip_viterbi_stream_create: device allocation failed");
  }
  *out = s;
  return SMILEHIP_OK;
}

extern "C" int smilehip_viterbi_stream_set_candidates(smilehip_viterbi_stream *s, int32_t n_candidates) {
  if (!s || n_candidates < 1 || n_candidates > 6 || s->n_frames != 0)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_viterbi_stream_set_candidates: 1 .. 6 candidates, before the first push");
  s->Q.n_cand = n_candidates;
  return SMILEHIP_OK;
}

static int viterbi_step(smilehip_viterbi_stream *s, int flush, int32_t *n_decided, int32_t *frames, int32_t *states, int32_t cap) {
  hipStream_t st = nullptr;                                  // a stream of single-frame steps: the null stream, synchronous
  hipError_t e = launch_f0_viterbi_step(s->Q, s->d_frames, s->d_st, s->d_dstate, s->d_paths, s->d_decided, flush, st);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "viterbi step launch failed: %s", hipGetErrorString(e));
  int h_st[4];
  int h_dec[2 * 128];
  if (hipMemcpyAsync(h_st, s->d_st, sizeof(h_st), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(h_dec, s->d_decided, sizeof(h_dec), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "viterbi step: copy back failed");
  const int n = h_st[3];
  if (n > cap || n > 128) return fail(SMILEHIP_ERR_INVALID, "viterbi step: %d decisions, room for %d", n, cap);
  for (int i = 0; i < n; ++i) { frames[i] = h_dec[2 * i]; states[i] = h_dec[2 * i + 1]; }
  *n_decided = n;
  return SMILEHIP_OK;
}

extern "C" int smilehip_viterbi_stream_push(smilehip_viterbi_stream *s, const float *cand_f0, const float *cand_voicing,
                                            int32_t *n_decided, int32_t *frames, int32_t *states, int32_t cap) {
  if (!s || !cand_f0 || !cand_voicing || !n_decided || !frames || !states) return fail(SMILEHIP_ERR_INVALID, "smilehip_viterbi_stream_push: null argument");
  if (s->n_frames == s->cap_frames) {                      // the device keeps every frame: transitions and decisions look back
    const int64_t ncap = s->cap_frames ? s->cap_frames * 2 : 4096;
    float *nf = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&nf), sizeof(float) * 21 * (size_t)ncap) != hipSuccess)
      return fail(SMILEHIP_ERR_HIP, "smilehip_viterbi_stream_push: device allocation failed");
    if (s->n_frames && hipMemcpy(nf, s->d_frames, sizeof(float) * 21 * (size_t)s->n_frames, hipMemcpyDeviceToDevice) != hipSuccess) {
      (void)hipFree(nf);
      return fail(SMILEHIP_ERR_HIP, "smilehip_viterbi_stream_push: copy failed");
    }
    if (s->d_frames) (void)hipFree(s->d_frames);
    s->d_frames = nf;
    s->cap_frames = ncap;
  }
  float row[21] = {0};
  for (int c = 0; c < s->Q.n_cand; ++c) { row[1 + c] = cand_f0[c]; row[7 + c] = cand_voicing[c]; }
  if (hipMemcpy(s->d_frames + 21 * s->n_frames, row, sizeof(row), hipMemcpyHostToDevice) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "smilehip_viterbi_stream_push: upload failed");
  s->n_frames++;
  return viterbi_step(s, 0, n_decided, frames, states, cap);
}

static int viterbi_reserve(smilehip_viterbi_stream *s, int64_t more) {     // the device keeps every frame: transitions and decisions look back
  if (s->n_frames + more <= s->cap_frames) return SMILEHIP_OK;
  int64_t ncap = s->cap_frames ? s->cap_frames * 2 : 4096;
  while (ncap < s->n_frames + more) ncap *= 2;
  float *nf = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&nf), sizeof(float) * 21 * (size_t)ncap) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "viterbi stream: device allocation failed");
  if (s->n_frames && hipMemcpy(nf, s->d_frames, sizeof(float) * 21 * (size_t)s->n_frames, hipMemcpyDeviceToDevice) != hipSuccess) {
    (void)hipFree(nf);
    return fail(SMILEHIP_ERR_HIP, "viterbi stream: copy failed");
  }
  if (s->d_frames) (void)hipFree(s->d_frames);
  s->d_frames = nf;
  s->cap_frames = ncap;
  return SMILEHIP_OK;
}

extern "C" int smilehip_viterbi_stream_push_frames(smilehip_viterbi_stream *s, const float *cand_f0, const float *cand_voicing, int64_t ld,
                                                   int32_t n_frames, int32_t *n_decided, int32_t *frames, int32_t *states, int32_t cap) {
  if (!s || !cand_f0 || !cand_voicing || !n_decided || !frames || !states || n_frames < 1 || ld < s->Q.n_cand)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_viterbi_stream_push_frames: bad argument");
  int rc = viterbi_reserve(s, n_frames);
  if (rc) return rc;
  const int64_t need = (int64_t)n_frames + f0_viterbi_max_buffer();
  if (need > s->cap_decided) {
    int *nd = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&nd), sizeof(int) * 2 * (size_t)need) != hipSuccess)
      return fail(SMILEHIP_ERR_HIP, "smilehip_viterbi_stream_push_frames: device allocation failed");
    (void)hipFree(s->d_decided);
    s->d_decided = nd;
    s->cap_decided = need;
  }
  std::vector<float> rows((size_t)n_frames * 21, 0.0f);
  for (int32_t f = 0; f < n_frames; ++f)
    for (int c = 0; c < s->Q.n_cand; ++c) {
      rows[(size_t)f * 21 + 1 + c] = cand_f0[(size_t)f * (size_t)ld + c];
      rows[(size_t)f * 21 + 7 + c] = cand_voicing[(size_t)f * (size_t)ld + c];
    }
  if (hipMemcpy(s->d_frames + 21 * s->n_frames, rows.data(), sizeof(float) * rows.size(), hipMemcpyHostToDevice) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "smilehip_viterbi_stream_push_frames: upload failed");
  s->n_frames += n_frames;
  hipStream_t st = nullptr;
  hipError_t e = launch_f0_viterbi_steps(s->Q, s->d_frames, s->d_st, s->d_dstate, s->d_paths, s->d_decided, n_frames, st);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "viterbi steps launch failed: %s", hipGetErrorString(e));
  int h_st[4];
  if (hipMemcpy(h_st, s->d_st, sizeof(h_st), hipMemcpyDeviceToHost) != hipSuccess) return fail(SMILEHIP_ERR_HIP, "viterbi steps: copy back failed");
  const int n = h_st[3];
  if (n > cap || n > need) return fail(SMILEHIP_ERR_INVALID, "viterbi steps: %d decisions, room for %d", n, cap);
  std::vector<int> h_dec((size_t)2 * (size_t)(n > 0 ? n : 1));
  if (n > 0 && hipMemcpy(h_dec.data(), s->d_decided, sizeof(int) * 2 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "viterbi steps: copy back failed");
  for (int i = 0; i < n; ++i) { frames[i] = h_dec[2 * (size_t)i]; states[i] = h_dec[2 * (size_t)i + 1]; }
  *n_decided = n;
  return SMILEHIP_OK;
}

extern "C" int smilehip_viterbi_stream_flush(smilehip_viterbi_stream *s, int32_t *n_decided, int32_t *frames, int32_t *states, int32_t cap) {
  if (!s || !n_decided || !frames || !states) return fail(SMILEHIP_ERR_INVALID, "smilehip_viterbi_stream_flush: null argument");
  if (s->n_frames == 0) { *n_decided = 0; return SMILEHIP_OK; }
  return viterbi_step(s, 1, n_decided, frames, states, cap);
}

// ---------------------------------------------- cPitchJitter as a stream (the plugin's tick-level override)
struct smilehip_jitter_stream {
  LldParams P;
  F0Params Q;
  int16_t *d_pcm = nullptr;
  int64_t cap_pcm = 0;
  float *d_f0 = nullptr, *d_out4 = nullptr, *d_shim = nullptr;
  int64_t cap_frames = 0, n_frames = 0;
  int64_t *d_off = nullptr;            // [0..1] frame_off, [2..3] samp_off
  double *d_state = nullptr;
};

extern "C" int smilehip_jitter_stream_destroy(smilehip_jitter_stream *s) {
  if (!s) return SMILEHIP_OK;
  if (s->d_pcm) (void)hipFree(s->d_pcm);
  if (s->d_f0) (void)hipFree(s->d_f0);
  if (s->d_out4) (void)hipFree(s->d_out4);
  if (s->d_shim) (void)hipFree(s->d_shim);
  if (s->d_off) (void)hipFree(s->d_off);
  if (s->d_state) (void)hipFree(s->d_state);
  delete s;
  return SMILEHIP_OK;
}

extern "C" int smilehip_jitter_stream_create(smilehip_context *ctx, double sample_period, int64_t frame_size, int64_t frame_step,
                                             double frame_step_sec, double search_range_rel, int32_t broken_jitter_thresh,
                                             smilehip_jitter_stream **out) {
  if (!ctx || !out || sample_period <= 0 || frame_size < 2 || frame_step < 1 || frame_step_sec <= 0 || search_range_rel <= 0)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_jitter_stream_create: bad argument");
  smilehip_jitter_stream *s = new smilehip_jitter_stream;
  std::memset(&s->P, 0, sizeof(s->P));
  std::memset(&s->Q, 0, sizeof(s->Q));
  s->P.n_utt = 1;
  s->Q.N = (int32_t)frame_size; s->Q.H = (int32_t)frame_step;
  s->Q.jit_Tw = sample_period;
  s->Q.jit_step_sec = frame_step_sec;
  s->Q.jit_search_range = search_range_rel;
  s->Q.jit_broken_thresh = broken_jitter_thresh;
  const double st0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMalloc(reinterpret_cast<void **>(&s->d_off), sizeof(int64_t) * 4) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&s->d_state), sizeof(st0)) != hipSuccess ||
      hipMemcpy(s->d_state, st0, sizeof(st0), hipMemcpyHostToDevice) != hipSuccess) {
    smilehip_jitter_stream_destroy(s);
    return fail(SMILEHIP_ERR_HIP, "smilehip_jitter_stream_create: device allocation failed");
  }
  *out = s;
  return SMILEHIP_OK;
}

template <typename T>
static int grow(T *&p, int64_t &cap, int64_t need, int64_t keep, int width) {
  if (need <= cap) return SMILEHIP_OK;
  int64_t ncap = cap ? cap : 4096;
  while (ncap < need) ncap *= 2;
  T *np = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&np), sizeof(T) * (size_t)ncap * width) != hipSuccess ||
      hipMemset(np, 0, sizeof(T) * (size_t)ncap * width) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "stream buffer allocation failed");
  if (p && keep > 0 && hipMemcpy(np, p, sizeof(T) * (size_t)keep * width, hipMemcpyDeviceToDevice) != hipSuccess) {
    (void)hipFree(np);
    return fail(SMILEHIP_ERR_HIP, "stream buffer copy failed");
  }
  if (p) (void)hipFree(p);
  p = np;
  cap = ncap;
  return SMILEHIP_OK;
}

extern "C" int smilehip_jitter_stream_set_time_offset(smilehip_jitter_stream *s, int64_t frames) {
  if (!s || frames < 0 || s->n_frames != 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_jitter_stream_set_time_offset: before the first push, frames >= 0");
  s->Q.jit_t_shift = frames;
  return SMILEHIP_OK;
}

extern "C" int smilehip_jitter_stream_push(smilehip_jitter_stream *s, float f0, const int16_t *h_pcm, int64_t pcm_start, int64_t n_pcm,
                                           float *out5, int64_t *last_idx, int64_t *last_mis) {
  if (!s || !out5 || n_pcm < 0 || pcm_start < 0 || (n_pcm > 0 && !h_pcm)) return fail(SMILEHIP_ERR_INVALID, "smilehip_jitter_stream_push: bad argument");
  int rc;
  const int64_t t = s->n_frames;
  {                                                         // the three per-frame arrays grow together
    int64_t c1 = s->cap_frames, c2 = s->cap_frames, c3 = s->cap_frames;
    if ((rc = grow(s->d_f0, c1, t + 1, t, 1)) || (rc = grow(s->d_out4, c2, t + 1, t, 4)) || (rc = grow(s->d_shim, c3, t + 1, t, 1))) return rc;
    s->cap_frames = c1;
  }
  const int64_t n_samp = n_pcm > 0 ? pcm_start + n_pcm : 0;           // no samples: the frame cannot be read (what the reference's
  if (n_pcm > 0) {                                                     // NULL matrix means); the state still moves on
    if ((rc = grow(s->d_pcm, s->cap_pcm, n_samp + 2, s->cap_pcm, 1))) return rc;
    if (hipMemcpy(s->d_pcm + pcm_start, h_pcm, sizeof(int16_t) * (size_t)n_pcm, hipMemcpyHostToDevice) != hipSuccess)
      return fail(SMILEHIP_ERR_HIP, "smilehip_jitter_stream_push: upload failed");
  } else if ((rc = grow(s->d_pcm, s->cap_pcm, 2, s->cap_pcm, 1))) return rc;
  const int64_t off[4] = {0, t + 1, 0, n_samp};
  if (hipMemcpy(s->d_off, off, sizeof(off), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->d_f0 + t, &f0, sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "smilehip_jitter_stream_push: upload failed");
  s->P.frame_off = s->d_off;
  s->P.samp_off = s->d_off + 2;
  s->P.pcm = s->d_pcm;
  s->P.total_frames = t + 1;
  s->Q.jit_shim_db = s->d_shim;
  s->Q.jit_stream = s->d_state;
  hipError_t e = launch_f0_jitter(s->P, s->Q, s->d_f0, 1, s->d_out4, nullptr);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "jitter step launch failed: %s", hipGetErrorString(e));
  double st[8];
  if (hipMemcpy(out5, s->d_out4 + 4 * t, sizeof(float) * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(out5 + 4, s->d_shim + t, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(st, s->d_state, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "smilehip_jitter_stream_push: copy back failed");
  if (last_idx) *last_idx = (int64_t)st[0];
  if (last_mis) *last_mis = (int64_t)st[1];
  s->n_frames = t + 1;
  return SMILEHIP_OK;
}

extern "C" int smilehip_jitter_stream_push_frames(smilehip_jitter_stream *s, const float *f0, int32_t n_frames, const int16_t *h_pcm,
                                                  int64_t pcm_start, int64_t n_pcm, float *out5, int64_t *last_idx, int64_t *last_mis) {
  if (!s || !f0 || !out5 || n_frames < 1 || n_pcm < 0 || pcm_start < 0 || (n_pcm > 0 && !h_pcm))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_jitter_stream_push_frames: bad argument");
  int rc;
  const int64_t t = s->n_frames;
  {
    int64_t c1 = s->cap_frames, c2 = s->cap_frames, c3 = s->cap_frames;
    if ((rc = grow(s->d_f0, c1, t + n_frames, t, 1)) || (rc = grow(s->d_out4, c2, t + n_frames, t, 4)) || (rc = grow(s->d_shim, c3, t + n_frames, t, 1))) return rc;
    s->cap_frames = c1;
  }
  const int64_t n_samp = pcm_start + n_pcm;                 // samples [0, n_samp) of the stream exist (the earlier ones are on the device)
  if ((rc = grow(s->d_pcm, s->cap_pcm, n_samp + 2, s->cap_pcm, 1))) return rc;
  if (n_pcm > 0 && hipMemcpy(s->d_pcm + pcm_start, h_pcm, sizeof(int16_t) * (size_t)n_pcm, hipMemcpyHostToDevice) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "smilehip_jitter_stream_push_frames: upload failed");
  const int64_t off[4] = {0, t + n_frames, 0, n_samp};
  if (hipMemcpy(s->d_off, off, sizeof(off), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->d_f0 + t, f0, sizeof(float) * (size_t)n_frames, hipMemcpyHostToDevice) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "smilehip_jitter_stream_push_frames: upload failed");
  s->P.frame_off = s->d_off;
  s->P.samp_off = s->d_off + 2;
  s->P.pcm = s->d_pcm;
  s->P.total_frames = t + n_frames;
  s->Q.jit_shim_db = s->d_shim;
  s->Q.jit_stream = s->d_state;
  hipError_t e = launch_f0_jitter(s->P, s->Q, s->d_f0, 1, s->d_out4, nullptr);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "jitter steps launch failed: %s", hipGetErrorString(e));
  std::vector<float> o4((size_t)n_frames * 4), sh((size_t)n_frames);
  double st[8];
  if (hipMemcpy(o4.data(), s->d_out4 + 4 * t, sizeof(float) * o4.size(), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(sh.data(), s->d_shim + t, sizeof(float) * sh.size(), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(st, s->d_state, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(SMILEHIP_ERR_HIP, "smilehip_jitter_stream_push_frames: copy back failed");
  for (int32_t f = 0; f < n_frames; ++f) {
    for (int k = 0; k < 4; ++k) out5[(size_t)f * 5 + k] = o4[(size_t)f * 4 + k];
    out5[(size_t)f * 5 + 4] = sh[(size_t)f];
  }
  if (last_idx) *last_idx = (int64_t)st[0];
  if (last_mis) *last_mis = (int64_t)st[1];
  s->n_frames = t + n_frames;
  return SMILEHIP_OK;
}

// ---------------------------------------------- the components the other INTERSPEECH sets add (lld_stage4_kernels.hip)
extern "C" int smilehip_intensity_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int32_t flags, float *d_dst,
                                         int64_t ld_dst, int64_t n_frames, void *stream) {
  const int n_out = ((flags & 1) ? 1 : 0) + ((flags & 2) ? 1 : 0);
  if (!ctx || N < 1 || N > (1 << 22) || !(flags & 3) || (flags & ~3) || n_frames < 0 || ld_src < (N < n_out ? N : n_out) || ld_dst < n_out ||
      (n_frames > 0 && (!d_src || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_intensity_frames: bad argument (flags: 1 intensity, 2 loudness)");
  // setupNamesForField (intensity.cpp:91-112): the Hamming window as doubles (smileUtil.c:1291-1303) and its sum in index order --
  // once per frame length (the plugin calls with one frame at a time)
  static std::mutex mu;
  static std::map<int64_t, std::array<double, 3>> cache;
  double w01[2] = {0.0, 0.0}, sum = 0.0;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(N);
    if (it == cache.end()) {
      const double NN = (double)N;
      long j = 0;
      for (double i = 0.0; i < NN; i += 1.0, ++j) {
        const double w = 0.54 - 0.46 * std::cos((2.0 * M_PI * i) / (NN - 1.0));
        if (j < 2) w01[j] = w;
        sum += w;
      }
      if (sum <= 0.0) sum = 1.0;
      it = cache.emplace(N, std::array<double, 3>{w01[0], w01[1], sum}).first;
    }
    w01[0] = it->second[0]; w01[1] = it->second[1]; sum = it->second[2];
  }
  const int n_sum = (int)(N < n_out ? N : n_out);              // MIN(Nsrc, MIN(nWin, Ndst)), :134
  STAGE_RET(stage_intensity(d_src, ld_src, n_sum, w01[0], w01[1], sum, flags, d_dst, ld_dst, n_frames, (hipStream_t)stream), "intensity");
}

extern "C" int smilehip_lsp_frames(smilehip_context *ctx, const float *d_lpc, int64_t ld_src, int32_t p, float *d_dst, int64_t ld_dst,
                                   int64_t n_frames, void *stream) {
  if (!ctx || p < 2 || p > 32 || n_frames < 0 || ld_src < p || ld_dst < p || (n_frames > 0 && (!d_lpc || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_lsp_frames: bad argument (2 <= p <= 32)");
  STAGE_RET(stage_lsp(d_lpc, ld_src, p, d_dst, ld_dst, n_frames, (hipStream_t)stream), "lsp");
}

extern "C" int smilehip_vecop_frames(smilehip_context *ctx, int32_t op, float param1, float logfloor, const float *d_src, int64_t ld_src,
                                     int32_t n_cols, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  const bool reduce = op >= SMILEHIP_VOP_X_SUM && op <= SMILEHIP_VOP_X_L2;
  if (!ctx || op < 0 || op > SMILEHIP_VOP_X_L2 || n_cols < 1 || n_frames < 0 || ld_src < n_cols || ld_dst < (reduce ? 1 : n_cols) ||
      (n_frames > 0 && (!d_src || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_vecop_frames: bad argument (op: SMILEHIP_VOP_*)");
  if (!(logfloor > 0.0f)) logfloor = (float)0.000000000001;            // vectorOperation.cpp:219-223
  float aux = param1;
  if (op == SMILEHIP_VOP_LOGA) {                                         // :99-107
    if (param1 <= 0.0f || param1 == 1.0f) param1 = (float)std::exp(1.0);
    aux = std::log(param1);
  }
  if (op == SMILEHIP_VOP_DB_POW) aux = (float)(10.0 / std::log(10.0));
  if (op == SMILEHIP_VOP_DB_MAG) aux = (float)(20.0 / std::log(10.0));
  STAGE_RET(stage_vecop(op, aux, logfloor, d_src, ld_src, n_cols, d_dst, ld_dst, n_frames, (hipStream_t)stream), "vecop");
}

extern "C" int smilehip_pitch_smoother_rows(smilehip_context *ctx, int32_t n_cand, float voicing_cutoff, int32_t octave_correction,
                                            int32_t post_simple, int32_t flags, const float *d_src, int64_t ld_src, const int64_t *d_row_off,
                                            int32_t n_streams, int64_t n_rows_single, void *d_state, int32_t resume, float *d_dst,
                                            int64_t ld_dst, int64_t *d_written, void *stream) {
  const int n_out = ((flags & 1) ? 1 : 0) + ((flags & 2) ? 1 : 0) + ((flags & 4) ? 1 : 0) + ((flags & 8) ? 1 : 0);
  if (!ctx || n_cand < 1 || n_cand > 16 || !(flags & 15) || (flags & ~15) || n_streams < 0 || ld_src < 3 * (int64_t)n_cand || ld_dst < n_out ||
      (!d_row_off && (n_streams > 1 || n_rows_single < 0)) || (n_streams > 0 && (!d_src || !d_dst)) || (resume && !d_state))
    return fail(SMILEHIP_ERR_INVALID,
                "smilehip_pitch_smoother_rows: bad argument (flags: 1 F0final, 2 F0finEnv, 4 voicingFinalClipped, 8 voicingFinalUnclipped; "
                "rows = [F0Cand | candVoicing | candScore], n_cand <= 16)");
  STAGE_RET(stage_pitch_smoother(n_cand, voicing_cutoff, octave_correction != 0, post_simple != 0, flags, d_src, ld_src, d_row_off, n_streams,
                                 n_rows_single, d_state, resume, d_dst, ld_dst, d_written, (hipStream_t)stream), "pitch smoother");
}

extern "C" int smilehip_specresample_geometry(int64_t n_in, double fs_sec, double last_fs_sec, double base_period, double target_fs,
                                              int64_t *n_out, int64_t *k_max, double *nd_out) {
  if (n_in < 2 || !(base_period > 0.0) || !(target_fs > 0.0) || !n_out || !k_max)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_specresample_geometry: bad argument");
  // cSpecResample::setupNewNames (specResample.cpp:117-172), resampleRatio = targetFs / sr (:94-103)
  const double sr = 1.0 / base_period;
  double ratio = target_fs / sr, nd;
  long no;
  if ((fs_sec != last_fs_sec) && (last_fs_sec != 0.0) && (last_fs_sec != base_period)) {
    const double nout0 = std::round((double)n_in * ratio * last_fs_sec / fs_sec);
    const double new_ratio = nout0 / ((double)n_in * (last_fs_sec / fs_sec));
    no = (long)nout0;
    if (new_ratio != ratio) ratio = new_ratio;
    nd = (double)n_in * ratio;
  } else {
    const double nout0 = std::round((double)n_in * ratio);
    no = (long)nout0;
    nd = nout0;
  }
  if (no < 1) return fail(SMILEHIP_ERR_INVALID, "smilehip_specresample_geometry: no output samples");
  long km = n_in > no ? no : (long)n_in;                                // antiAlias = 1 (smileDsp_initIrdft, smileUtil.c:1752-1786)
  if (km & 1) km--;
  *n_out = no;
  *k_max = km;
  if (nd_out) *nd_out = nd;
  return SMILEHIP_OK;
}

extern "C" int smilehip_specresample_tables(int64_t n_in, int64_t n_out, int64_t k_max, double nd, float *cos_table, float *sin_table) {
  if (n_in < 2 || n_out < 1 || k_max < 0 || (k_max & 1) || k_max > n_in || !(nd > 0.0) || !cos_table || !sin_table)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_specresample_tables: bad argument");
  // up-sampling adds the Nyquist term, kept in the last slot of a table row: the row must be the full half spectrum
  if (n_out >= n_in && k_max != (n_in & ~(int64_t)1))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_specresample_tables: n_out >= n_in needs k_max = n_in (even) / n_in - 1 (odd), got %lld", (long long)k_max);
  const int64_t h = k_max / 2;
  const double pi2 = 2.0 * M_PI;
  for (int64_t i = 0; i < h * n_out; ++i) cos_table[i] = sin_table[i] = 0.0f;
  for (int64_t i = 0; i < n_out; i++) {
    const int64_t i_n = i * h - 1;
    if (n_out >= n_in) cos_table[i_n + n_in / 2] = (float)std::cos((pi2 * (double)((n_in / 2) * i)) / nd);
    for (int64_t k = 2; k < k_max; k += 2) {
      const double kn = pi2 * (double)(k / 2 * i) / nd;
      cos_table[i_n + k / 2] = (float)std::cos(kn);
      sin_table[i_n + k / 2] = (float)std::sin(kn);
    }
  }
  return SMILEHIP_OK;
}

extern "C" int smilehip_specresample_table_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t n_in, int64_t n_out,
                                                  int64_t k_max, const float *d_cos, const float *d_sin, float *d_dst, int64_t ld_dst,
                                                  int64_t n_frames, void *stream) {
  if (!ctx || n_in < 2 || n_in > 8192 || n_out < 1 || n_out > (1 << 20) || k_max < 2 || (k_max & 1) || k_max > n_in || n_frames < 0 ||
      ld_src < n_in || ld_dst < n_out || !d_cos || !d_sin || (n_frames > 0 && (!d_src || !d_dst)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_specresample_table_frames: bad argument (n_in <= 8192; tables of smilehip_specresample_tables)");
  if (n_out >= n_in && k_max != (n_in & ~(int64_t)1))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_specresample_table_frames: n_out >= n_in needs k_max = n_in (even) / n_in - 1 (odd), got %lld", (long long)k_max);
  STAGE_RET(stage_specresample_g(d_src, ld_src, (int)n_in, (int)n_out, (int)k_max, d_cos, d_sin, d_dst, ld_dst, n_frames, (hipStream_t)stream),
            "specresample");
}

extern "C" int smilehip_lpc_acf_frames(smilehip_context *ctx, const float *d_x, int64_t ld_src, int64_t n, int32_t p, float *d_lpc,
                                       int64_t ld_dst, int64_t n_frames, void *stream) {
  if (!ctx || p < 1 || p > 32 || n <= p || n > 15000 || n_frames < 0 || ld_src < n || ld_dst < p || (n_frames > 0 && (!d_x || !d_lpc)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_lpc_acf_frames: bad argument (1 <= p <= 32 < n <= 15000)");
  STAGE_RET(stage_lpc_g(d_x, ld_src, (int)n, p, d_lpc, ld_dst, n_frames, (hipStream_t)stream), "lpc");
}
