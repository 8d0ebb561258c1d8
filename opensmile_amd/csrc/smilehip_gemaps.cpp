// libsmilehip, C ABI part 6: per-component operators of the GeMAPS / eGeMAPS LLD set (what the plugin's overrides call;
// kernels in lld_gemaps.hip). All take an eGeMAPS plan (smilehip_config_egemapsv02), whose tables fix the geometry.
#include "smilehip_internal.hpp"

namespace {
int need_egemaps(const smilehip_plan *p, const char *fn) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "%s: null plan", fn);
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (p->cfg.chain_kind != SMILEHIP_CHAIN_EGEMAPS) return fail(SMILEHIP_ERR_INVALID, "%s: needs an eGeMAPS plan (smilehip_config_egemapsv02)", fn);
  return SMILEHIP_OK;
}
int check_rows(const void *s, const void *d, int64_t lds, int64_t ldd, int64_t n, int64_t ws, int64_t wd, const char *fn) {
  if (n < 0 || (n > 0 && (!s || !d))) return fail(SMILEHIP_ERR_INVALID, "%s: null pointer", fn);
  if (lds < ws || ldd < wd) return fail(SMILEHIP_ERR_INVALID, "%s: leading dimension too small", fn);
  return SMILEHIP_OK;
}
}  // namespace

extern "C" int smilehip_spectral_gemaps_frames(smilehip_plan *p, const float *d_mag, int64_t ld_src, float *d_state, int first,
                                               float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  int rc = need_egemaps(p, "smilehip_spectral_gemaps_frames");
  if (rc) return rc;
  if (!d_state && n_frames > 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_gemaps_frames: null state buffer");
  if ((rc = check_rows(d_mag, d_dst, ld_src, ld_dst, n_frames, p->geo.K, 5, "smilehip_spectral_gemaps_frames"))) return rc;
  GemapsParams G;
  gemaps_plan_consts(p, G);
  hipError_t e = launch_gemaps_spectral_rows(d_mag, ld_src, d_state, first != 0, d_dst, ld_dst, n_frames, (int)p->geo.K, G, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "GeMAPS spectral kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_specresample_frames(smilehip_plan *p, const float *d_spec, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                            int64_t n_frames, void *stream) {
  int rc = need_egemaps(p, "smilehip_specresample_frames");
  if (rc) return rc;
  if ((rc = check_rows(d_spec, d_dst, ld_src, ld_dst, n_frames, p->geo.Nfft, 220, "smilehip_specresample_frames"))) return rc;
  GemapsParams G;
  gemaps_plan_consts(p, G);
  G.op_mode = 1; G.op_in = d_spec; G.op_ld_in = ld_src; G.op_out = d_dst; G.op_ld_out = ld_dst; G.op_rows = n_frames;
  hipError_t e = launch_gemaps_lpc_rows(G, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "cSpecResample kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_lpc_frames(smilehip_plan *p, const float *d_x, int64_t ld_src, float *d_lpc, int64_t ld_dst, int64_t n_frames,
                                   void *stream) {
  int rc = need_egemaps(p, "smilehip_lpc_frames");
  if (rc) return rc;
  if ((rc = check_rows(d_x, d_lpc, ld_src, ld_dst, n_frames, 220, 11, "smilehip_lpc_frames"))) return rc;
  GemapsParams G;
  gemaps_plan_consts(p, G);
  G.op_mode = 2; G.op_in = d_x; G.op_ld_in = ld_src; G.lpc = d_lpc; G.lpc_ld = ld_dst; G.op_rows = n_frames;
  hipError_t e = launch_gemaps_lpc_rows(G, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "cLpc kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_formantlpc_frames(smilehip_plan *p, const float *d_lpc, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                          int64_t n_frames, void *stream) {
  int rc = need_egemaps(p, "smilehip_formantlpc_frames");
  if (rc) return rc;
  if ((rc = check_rows(d_lpc, d_dst, ld_src, ld_dst, n_frames, 11, 10, "smilehip_formantlpc_frames"))) return rc;
  GemapsParams G;
  gemaps_plan_consts(p, G);
  G.op_mode = 1; G.lpc = const_cast<float *>(d_lpc); G.lpc_ld = ld_src; G.formants = d_dst; G.fm_ld = ld_dst; G.op_rows = n_frames;
  hipError_t e = launch_gemaps_formant_rows(G, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "cFormantLpc kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_harmonics_frames(smilehip_plan *p, const float *d_f0, const float *d_formants, int64_t ld_formants,
                                         const float *d_mag, int64_t ld_mag, float *d_dst, int64_t ld_dst, int64_t n_frames,
                                         void *stream) {
  int rc = need_egemaps(p, "smilehip_harmonics_frames");
  if (rc) return rc;
  if (n_frames > 0 && (!d_f0 || !d_formants)) return fail(SMILEHIP_ERR_INVALID, "smilehip_harmonics_frames: null pointer");
  if (ld_formants < 10) return fail(SMILEHIP_ERR_INVALID, "smilehip_harmonics_frames: ld_formants < 10");
  if ((rc = check_rows(d_mag, d_dst, ld_mag, ld_dst, n_frames, p->f0_plan->geo.K, 6, "smilehip_harmonics_frames"))) return rc;
  GemapsParams G;
  gemaps_plan_consts(p, G);
  G.op_mode = 1; G.op_f0 = d_f0; G.formants = const_cast<float *>(d_formants); G.fm_ld = ld_formants;
  G.op_in = d_mag; G.op_ld_in = ld_mag; G.op_out = d_dst; G.op_ld_out = ld_dst; G.op_rows = n_frames;
  LldParams P;
  std::memset(&P, 0, sizeof(P));
  F0Params Q;
  fill_f0_params(p->f0_plan, Q);
  hipError_t e = launch_gemaps_harm(P, Q, G, p->ctx->prop.multiProcessorCount, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "cHarmonics kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}
