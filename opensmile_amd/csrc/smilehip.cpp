// libsmilehip.so -- C ABI implementation (see include/smilehip.h).
// Host side: table generation (tables.cpp), device buffers, launch logic.
// There is no CPU fallback: every compute entry point launches HIP kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/smilehip.h"
#include "lld_launch.hpp"
#include "lld_params.hpp"
#include "tables.hpp"

using namespace smilehip;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(SMILEHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                   \
  } while (0)

extern "C" const char *smilehip_last_error(void) { return g_err.c_str(); }
extern "C" int smilehip_version(void) { return SMILEHIP_VERSION; }

// ----------------------------------------------------------------- objects
struct smilehip_context {
  int device = 0;
  hipDeviceProp_t prop{};
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  int upload(const std::vector<T> &h) {
    release();
    n = h.size();
    const size_t bytes = (n ? n : 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), bytes));
    if (n) HIP_TRY(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return SMILEHIP_OK;
  }
};

struct smilehip_plan {
  smilehip_context *ctx = nullptr;
  smilehip_lld_config cfg{};
  Geometry geo;
  std::vector<float> h_window;
  MelBank mel;
  DctTables dct;
  DevBuf<float> d_window, d_mel_coef, d_dct_rows, d_dct_gain;
  DevBuf<int32_t> d_mel_rng;
  DevBuf<float2> d_tw_half, d_tw_full, d_tw256, d_tw512, d_fwin;
  DevBuf<float4> d_melw;
  DevBuf<uint32_t> d_melo;
  DevBuf<float> d_dct28;
  DevBuf<int32_t> d_band_slots;
  Fast512Host fast;
  bool use_fast = false;
  DevBuf<float> d_eql, d_eql_log;
  DevBuf<float> d_plp_eql, d_plp_cos, d_plp_sin;     // PLP chain tables
  std::vector<float> h_plp_cos;
  DevBuf<double> d_sharp;
  float rasta_iir = 0.f, rasta_fir[5] = {0, 0, 0, 0, 0};
  int32_t band_iL[2] = {0, 0}, band_iR[2] = {0, 0};
  double band_wL[2] = {0, 0}, band_wR[2] = {0, 0}, slope_Sf = 0, slope_S2f = 0;
  // timing
  // HIP-event timing ring: slot i holds {before main, after main, after delta}
  static constexpr int kRing = 128;
  bool timing = false;
  hipEvent_t ev[kRing][3] = {};
  int64_t n_timed = 0;
  int force_generic = 0;
  uint32_t stage_mask = SMILEHIP_STAGE_ALL;
  ~smilehip_plan() {
    for (auto &slot : ev)
      for (auto &e : slot)
        if (e) (void)hipEventDestroy(e);
  }
};

struct smilehip_batch {
  smilehip_plan *plan = nullptr;
  int32_t n_utt = 0;
  int64_t total_frames = 0;
  std::vector<int64_t> h_samp_off, h_frame_off, h_row_off;
  int64_t total_rows = 0;
  DevBuf<int64_t> d_row_off;
  DevBuf<float> d_raw16;        // IS09: pre-smoothing LLD columns, total_frames x 16
  DevBuf<float> d_static;       // MFCC chain with deltas: compact static block, total_frames x n_mfcc
  DevBuf<float> d_rawA, d_rawB, d_mel1;   // ComParE A+B scratch
  DevBuf<int32_t> d_run_utt, d_run_t0;
  int32_t n_runs = 0;
  std::vector<int32_t> h_short;
  DevBuf<int64_t> d_samp_off, d_frame_off;
  DevBuf<int32_t> d_tile_utt, d_tile_t0, d_short, d_dtile_utt, d_dtile_t0;
  DevBuf<TileRec> d_tile_rec;
  int32_t n_tiles = 0, n_dtiles = 0;
  bool all_even = true;      // every utterance with frames starts at an even sample offset
};

// ------------------------------------------------------------- life cycle
extern "C" int smilehip_init(int device, smilehip_context **out) {
  if (!out) return fail(SMILEHIP_ERR_INVALID, "smilehip_init: null output pointer");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(SMILEHIP_ERR_NO_DEVICE, "no HIP device visible (libsmilehip has no CPU fallback)");
  if (device < 0 || device >= n) return fail(SMILEHIP_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
  auto *c = new (std::nothrow) smilehip_context();
  if (!c) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
    delete c;
    return fail(SMILEHIP_ERR_HIP, "cannot open HIP device %d", device);
  }
  if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
    std::string arch = c->prop.gcnArchName;
    delete c;
    return fail(SMILEHIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                arch.c_str());
  }
  *out = c;
  return SMILEHIP_OK;
}

extern "C" void smilehip_shutdown(smilehip_context *ctx) { delete ctx; }

extern "C" int smilehip_device_name(smilehip_context *ctx, char *buf, int buflen) {
  if (!ctx || !buf || buflen <= 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_device_name: bad argument");
  snprintf(buf, buflen, "%s (%s, %d CUs)", ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.multiProcessorCount);
  return SMILEHIP_OK;
}

extern "C" int smilehip_alloc(smilehip_context *ctx, uint64_t bytes, void **d_ptr) {
  if (!ctx || !d_ptr) return fail(SMILEHIP_ERR_INVALID, "smilehip_alloc: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
  return SMILEHIP_OK;
}
extern "C" int smilehip_free(smilehip_context *ctx, void *d_ptr) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_free: null context");
  if (d_ptr) HIP_TRY(hipFree(d_ptr));
  return SMILEHIP_OK;
}
extern "C" int smilehip_copy_to_device(smilehip_context *ctx, void *d_dst, const void *h_src, uint64_t bytes, void *stream) {
  if (!ctx || (bytes && (!d_dst || !h_src))) return fail(SMILEHIP_ERR_INVALID, "smilehip_copy_to_device: null argument");
  if (bytes) HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return SMILEHIP_OK;
}
extern "C" int smilehip_copy_to_host(smilehip_context *ctx, void *h_dst, const void *d_src, uint64_t bytes, void *stream) {
  if (!ctx || (bytes && (!h_dst || !d_src))) return fail(SMILEHIP_ERR_INVALID, "smilehip_copy_to_host: null argument");
  if (bytes) HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return SMILEHIP_OK;
}
extern "C" int smilehip_stream_synchronize(smilehip_context *ctx, void *stream) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_stream_synchronize: null context");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SMILEHIP_OK;
}

extern "C" void smilehip_config_mfcc12_0_d_a(smilehip_lld_config *c) {
  std::memset(c, 0, sizeof(*c));
  c->struct_size = sizeof(*c);
  c->sample_rate = 16000.0;
  c->frame_size_sec = 0.0250;
  c->frame_step_sec = 0.010;
  c->preemph = 1;
  c->preemph_k = 0.97f;
  c->preemph_de = 0;
  c->win_func = SMILEHIP_WIN_HAMM;
  c->win_sigma = 0.4;
  c->win_gain = 1.0;
  c->win_offset = 0.0;
  c->zero_pad_symmetric = 0;
  c->n_bands = 26;
  c->lofreq = 0.0f;
  c->hifreq = 8000.0f;
  c->use_power = 1;
  c->mel_htk_compatible = 1;
  c->first_mfcc = 0;
  c->last_mfcc = 12;
  c->cep_lifter = 22.0f;
  c->mfcc_htk_compatible = 1;
  c->melfloor = 1e-8f;
  c->n_delta = 2;
  c->delta_win = 2;
}

extern "C" void smilehip_config_is09_lld(smilehip_lld_config *c) {
  smilehip_config_mfcc12_0_d_a(c);
  c->chain_kind = SMILEHIP_CHAIN_IS09;
  c->use_power = 0;            // [is09_mspec] usePower = 0
  c->first_mfcc = 1;
  c->last_mfcc = 12;
  c->n_delta = 1;
  c->delta_win = 2;
  c->pitch_max = 500.0;
  c->voicing_cutoff = 0.55;
  c->sma_win = 3;
}

extern "C" void smilehip_config_plp_0_d_a(smilehip_lld_config *c) {
  smilehip_config_mfcc12_0_d_a(c);        // same front end: 25 ms / 10 ms, k = 0.97, ham, 26 HTK mel bands
  c->chain_kind = SMILEHIP_CHAIN_PLP;
  c->plp_lp_order = 5;                    // [plp:cPlp] lpOrder = 5, compression = 0.33, cepLifter = 22
  c->plp_compression = 0.33f;
  c->cep_lifter = 22.0f;
  c->first_mfcc = 0;
  c->last_mfcc = 5;
}

extern "C" void smilehip_config_compare16_ab(smilehip_lld_config *c) {
  smilehip_config_mfcc12_0_d_a(c);
  c->chain_kind = SMILEHIP_CHAIN_COMPARE_AB;
  c->frame_size_sec = 0.020;       // [is13_frame25]
  c->preemph = 0;
  c->win_func = SMILEHIP_WIN_HAMM; // [is13_win25]
  c->zero_pad_symmetric = 1;       // [is13_fft25]
  c->lofreq = 20.0f;               // [is13_melspec1], [is13_melspecMfcc]
  c->use_power = 1;
  c->first_mfcc = 1;               // [is13_mfcc]
  c->last_mfcc = 14;
  c->n_delta = 1;
  c->delta_win = 2;
  c->sma_win = 3;
}

// ------------------------------------------------------------------- plan
static int build_tables(smilehip_plan *p, bool upload = true) {
  int rc;
  if ((rc = make_geometry(p->cfg, p->geo)) != SMILEHIP_OK) return fail(rc, "invalid framing parameters");
  if (p->geo.Nfft > 8192) return fail(SMILEHIP_ERR_INVALID, "FFT length %lld > 8192 unsupported", (long long)p->geo.Nfft);
  const uint32_t mask = p->cfg.stage_mask ? p->cfg.stage_mask : SMILEHIP_STAGE_ALL;
  p->h_window.assign(size_t(p->geo.N), 1.0f);
  if ((mask & SMILEHIP_STAGE_WINDOW) && (rc = make_window(p->cfg, p->geo.N, p->h_window)) != SMILEHIP_OK)
    return fail(rc, "unknown window function %d", p->cfg.win_func);
  if (mask & SMILEHIP_STAGE_MEL) {
    if ((rc = make_mel(p->cfg, p->geo, p->mel)) != SMILEHIP_OK) return fail(rc, "invalid mel bank parameters");
  } else {
    p->mel = MelBank();
    p->mel.n_bands = p->cfg.n_bands;
  }
  if (mask & SMILEHIP_STAGE_MFCC) {
    if ((rc = make_dct(p->cfg, p->dct)) != SMILEHIP_OK) return fail(rc, "invalid MFCC range");
  } else {
    p->dct = DctTables();
  }
  p->stage_mask = mask;
  if (p->cfg.n_delta < 0 || p->cfg.n_delta > 2) return fail(SMILEHIP_ERR_INVALID, "n_delta must be 0..2");
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_IS09) {
    if (mask != SMILEHIP_STAGE_ALL || p->dct.n_mfcc != 12 || p->cfg.n_delta != 1 || p->cfg.sma_win < 3 || !(p->cfg.sma_win & 1) ||
        p->cfg.sma_win > 9)
      return fail(SMILEHIP_ERR_INVALID, "IS09 chain needs 12 MFCC, one delta stage and an odd smaWin in 3..9");
  } else if (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB) {
    if (mask != SMILEHIP_STAGE_ALL || p->dct.n_mfcc != 14 || p->mel.n_bands != 26 || p->cfg.n_delta != 1 ||
        p->cfg.delta_win != 2 || p->cfg.sma_win != 3 || !p->cfg.use_power || p->cfg.preemph || p->geo.Nfft != 512)
      return fail(SMILEHIP_ERR_INVALID, "ComParE A+B chain: unsupported parameter set");
  } else if (p->cfg.chain_kind == SMILEHIP_CHAIN_PLP) {
    if (mask != SMILEHIP_STAGE_ALL || p->cfg.plp_lp_order < 1 || p->cfg.plp_lp_order > 15 || !p->cfg.mel_htk_compatible ||
        !p->cfg.use_power || p->mel.n_bands > 30 || p->cfg.plp_compression < 0.0f)
      return fail(SMILEHIP_ERR_INVALID, "PLP chain: lpOrder 1..15, HTK-scaled power mel bands (<= 30) required");
    p->dct.n_mfcc = p->cfg.plp_lp_order + 1;            // outputs of the chain's static block
    p->dct.melfloor = 1.0f;                              // htkcompatible forces melfloor = 1.0 (plp.cpp:150-160)
  } else if (p->cfg.chain_kind != SMILEHIP_CHAIN_MFCC) {
    return fail(SMILEHIP_ERR_INVALID, "unknown chain_kind %d", p->cfg.chain_kind);
  }
  if (p->cfg.n_delta > 0 && (p->cfg.delta_win < 1 || p->cfg.delta_win > 4))
    return fail(SMILEHIP_ERR_INVALID, "delta_win must be 1..4");

  const int64_t M = p->geo.Nfft / 2;
  std::vector<float2> twh(static_cast<size_t>(M / 2 > 0 ? M / 2 : 1)), twf(static_cast<size_t>(M / 2 + 1));
  for (int64_t j = 0; j < M / 2; ++j) {
    const double a = -2.0 * M_PI * double(j) / double(M);
    twh[j] = make_float2(float(std::cos(a)), float(std::sin(a)));
  }
  for (int64_t k = 0; k <= M / 2; ++k) {
    const double a = -2.0 * M_PI * double(k) / double(p->geo.Nfft);
    twf[k] = make_float2(float(std::cos(a)), float(std::sin(a)));
  }
  std::vector<int32_t> rng((mask & SMILEHIP_STAGE_MEL) ? size_t(4) * p->mel.n_bands : 0);
  for (int b = 0; b < p->mel.n_bands && (mask & SMILEHIP_STAGE_MEL); ++b) {
    rng[4 * b + 0] = p->mel.rise_lo[b];
    rng[4 * b + 1] = p->mel.rise_hi[b];
    rng[4 * b + 2] = p->mel.fall_lo[b];
    rng[4 * b + 3] = p->mel.fall_hi[b];
  }
  // PLP chain: cPlp::initTables (plp.cpp:288-357): IDFT cosine table, lifter table, HTK equal-loudness weights at
  // the band centres cMelspec publishes (melspec.cpp:408-412)
  std::vector<float> plp_eql(32, 0.0f), plp_sin(16, 1.0f);
  const bool is_plp = p->cfg.chain_kind == SMILEHIP_CHAIN_PLP;
  if (is_plp) {
    const int nB = p->mel.n_bands, nFreq = nB + 2, nAuto = p->cfg.plp_lp_order + 1;
    p->h_plp_cos.assign(size_t(nAuto) * nFreq, 0.0f);
    const float a = (float)M_PI / (float)(nFreq - 1);
    for (int i = 0; i < nAuto; i++) {
      const int ib = i * nFreq;
      int m;
      p->h_plp_cos[ib] = 1.0f;
      for (m = 1; m < (nFreq - 1); m++) p->h_plp_cos[m + ib] = (float)(2.0 * std::cos(a * (double)i * (double)m));
      p->h_plp_cos[m + ib] = (float)(std::cos(a * (double)i * (double)m));
    }
    const float L = (float)(int)p->cfg.cep_lifter;      // cepLifter is read with getInt (plp.cpp:142)
    for (int i = 0; i < nAuto; i++)
      plp_sin[i] = (L > 0.0f) ? ((float)1.0 + L / (float)2.0 * std::sin((float)M_PI * ((float)(i)) / L)) : 1.0f;
    for (int m = 1; m <= nB; ++m) {
      const double hz = 700.0 * (std::exp(double(p->mel.centres[m]) / 1127.0) - 1.0);
      const double f2 = hz * hz, fs = f2 / (f2 + 1.6e5);
      plp_eql[m - 1] = (float)(fs * fs * ((f2 + 1.44e6) / (f2 + 9.61e6)));   // smileDsp_equalLoudnessWeight_htk
    }
  }
  // fast Nfft=512 kernel if the geometry allows it (SMILEHIP_FORCE_GENERIC=1 disables it)
  p->use_fast = false;
  if (mask == SMILEHIP_STAGE_ALL && (p->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || (is_plp && p->mel.n_bands == 26)) &&
      !p->force_generic && fast512_applicable((int)p->geo.Nfft, (int)p->geo.N)) {
    p->use_fast = fast512_build_host(p->cfg, p->geo, p->h_window, p->mel, p->dct, p->fast) == 0;
    if (p->use_fast && is_plp) {                          // the DCT rows' place holds the IDFT cosine rows (28 floats each)
      p->fast.dct28.assign(16 * 28, 0.0f);
      std::copy(p->h_plp_cos.begin(), p->h_plp_cos.end(), p->fast.dct28.begin());
    }
    if (p->ctx) p->fast.max_blocks = 2 * p->ctx->prop.multiProcessorCount;
  }
  if (!upload) return SMILEHIP_OK;
  if ((rc = p->d_window.upload(p->h_window))) return rc;
  if ((rc = p->d_mel_coef.upload(p->mel.coef))) return rc;
  if ((rc = p->d_mel_rng.upload(rng))) return rc;
  if ((rc = p->d_dct_rows.upload(p->dct.cos_rows))) return rc;
  if ((rc = p->d_dct_gain.upload(p->dct.gain))) return rc;
  if ((rc = p->d_tw_half.upload(twh))) return rc;
  if ((rc = p->d_tw_full.upload(twf))) return rc;
  if (is_plp && ((rc = p->d_plp_eql.upload(plp_eql)) || (rc = p->d_plp_cos.upload(p->h_plp_cos)) || (rc = p->d_plp_sin.upload(plp_sin))))
    return rc;
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB) {
    // cPlp::initTables (plp.cpp:335-402): equal-loudness weights at the band centres cMelspec
    // publishes as metadata (melspec.cpp:408-412), and the newRASTA filter coefficients
    std::vector<float> eql(26), eqll(26);
    for (int m = 1; m <= 26; ++m) {
      const double hz = 700.0 * (std::exp(double(p->mel.centres[m]) / 1127.0) - 1.0);
      const double w = 2.0 * M_PI * hz, w2 = w * w, c = w2 + 6300000.0;
      const double e = (c > 0.0) ? (1e32 * ((w2 + 56.8e6) * w2 * w2) / (c * c * (w2 + 0.38e9) * (w2 * w2 * w2 * w + 1.7e31))) : 0.0;
      eql[m - 1] = float(e);
      eqll[m - 1] = std::log(eql[m - 1]);
    }
    const double Tl = p->geo.frame_period;
    const float lower = 1.0f, upper = 29.0f;
    p->rasta_iir = float(1.0 - std::sin(2.0 * M_PI * lower * Tl));
    const float om = float(std::cos(2.0 * M_PI * upper * Tl));
    const float norm = float(std::sqrt(10.0 * (32.0 * om * om + 8.0)));
    p->rasta_fir[0] = float(2.0 / norm);
    p->rasta_fir[1] = float(-4.0 * om / norm);
    p->rasta_fir[2] = 0.0f;
    p->rasta_fir[3] = -p->rasta_fir[1];
    p->rasta_fir[4] = -p->rasta_fir[0];
    if ((rc = p->d_eql.upload(eql)) || (rc = p->d_eql_log.upload(eqll))) return rc;
  }
  if (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB || (mask & SMILEHIP_STAGE_SPECTRAL)) {
    // sharpness weights (spectral.cpp:1440-1455): bark(f) * g(bark(f)) for bins 1..K-1
    std::vector<double> sw(size_t(p->geo.K - 1));
    const double F0 = 1.0 / p->geo.fft_frame_size_sec;
    for (int64_t j = 1; j < p->geo.K; ++j) {
      const double x = F0 * double(j);
      double zz = 0.0;
      if (x > 0) {
        zz = (26.81 / (1.0 + 1960.0 / x)) - 0.53;
        if (zz < 2) zz = 0.85 * zz + 0.3;
        else if (zz > 20.1) zz = 1.22 * zz - 0.22 * 20.1;
      }
      const double g = (zz <= 16.0) ? 1.0 : std::pow((zz - 16.0) / 4.0, 1.5849625) + 1.0;
      sw[size_t(j - 1)] = zz * g;
    }
    // band edges of [is13_spectral] bands 250-650 and 1000-4000 (spectral.cpp:779-853)
    const int band_lo[2] = {250, 1000}, band_hi[2] = {650, 4000};
    const int Nsrc = (int)p->geo.K;
    for (int b = 0; b < 2; ++b) {
      int ii;
      double wghtL, wghtR, idxL, idxR;
      for (ii = 0; ii < Nsrc; ii++) if (F0 * ii > (double)band_lo[b]) break;
      if ((ii < Nsrc) && (ii > 0)) wghtL = (F0 * ii - (double)band_lo[b]) / (F0 * ii - F0 * (ii - 1)); else wghtL = 1.0;
      idxL = (double)ii - 1.0;
      if (idxL < 0) idxL = 0;
      if (idxL >= Nsrc) idxL = Nsrc;
      if (wghtL == 0.0) wghtL = 1.0;
      for (ii = 0; ii < Nsrc; ii++) if (F0 * ii >= (float)band_hi[b]) break;
      if ((ii < Nsrc) && (ii > 0)) wghtR = ((double)band_hi[b] - F0 * (ii - 1)) / (F0 * ii - F0 * (ii - 1)); else wghtR = 1.0;
      if ((ii < Nsrc) && (F0 * ii == (float)band_hi[b])) idxR = (double)ii; else idxR = (double)ii - 1.0;
      if (idxR >= Nsrc) idxR = Nsrc - 1;
      if (wghtR == 0.0) wghtR = 1.0;
      int iL = (int)std::floor(idxL), iR = (int)std::floor(idxR);
      if (iL >= Nsrc) { iL = iR = Nsrc - 1; wghtR = 0.0; wghtL = 0.0; }
      if (iR >= Nsrc) { iR = Nsrc - 1; wghtR = 1.0; }
      if (iL < 0) iL = 0;
      if (iR < 0) iR = 0;
      p->band_iL[b] = iL; p->band_iR[b] = iR; p->band_wL[b] = wghtL; p->band_wR[b] = wghtR;
    }
    double Sf = 0.0, S2f = 0.0;
    for (int64_t i = 1; i < p->geo.K; ++i) { S2f += (F0 * i) * (F0 * i); Sf += F0 * i; }
    p->slope_Sf = Sf;
    p->slope_S2f = S2f;
    if ((rc = p->d_sharp.upload(sw))) return rc;
  }
  if (p->use_fast) {
    if ((rc = p->d_tw256.upload(p->fast.tw256))) return rc;
    if ((rc = p->d_tw512.upload(p->fast.tw512))) return rc;
    if ((rc = p->d_fwin.upload(p->fast.win))) return rc;
    if ((rc = p->d_melw.upload(p->fast.melw))) return rc;
    if ((rc = p->d_melo.upload(p->fast.melo))) return rc;
    if ((rc = p->d_dct28.upload(p->fast.dct28))) return rc;
    if ((rc = p->d_band_slots.upload(p->fast.band_slots))) return rc;
  }
  return SMILEHIP_OK;
}

extern "C" int smilehip_plan_create(smilehip_context *ctx, const smilehip_lld_config *cfg, smilehip_plan **out) {
  if (!ctx || !cfg || !out) return fail(SMILEHIP_ERR_INVALID, "smilehip_plan_create: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(smilehip_lld_config))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(smilehip_lld_config));
  HIP_TRY(hipSetDevice(ctx->device));
  auto *p = new (std::nothrow) smilehip_plan();
  if (!p) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  p->ctx = ctx;
  p->cfg = *cfg;
  const char *fg = getenv("SMILEHIP_FORCE_GENERIC");
  p->force_generic = (fg && fg[0] == '1') ? 1 : 0;
  int rc = build_tables(p);
  if (rc != SMILEHIP_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return SMILEHIP_OK;
}

extern "C" int smilehip_plan_create_host_only(const smilehip_lld_config *cfg, smilehip_plan **out) {
  if (!cfg || !out) return fail(SMILEHIP_ERR_INVALID, "smilehip_plan_create_host_only: null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(smilehip_lld_config))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(smilehip_lld_config));
  auto *p = new (std::nothrow) smilehip_plan();
  if (!p) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  p->cfg = *cfg;
  int rc = build_tables(p, false);
  if (rc != SMILEHIP_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return SMILEHIP_OK;
}

extern "C" void smilehip_plan_destroy(smilehip_plan *plan) { delete plan; }

static int plan_n_static(const smilehip_plan *p) {
  return p->cfg.chain_kind == SMILEHIP_CHAIN_IS09 ? 16 : (p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB ? 59 : p->dct.n_mfcc);
}
static int plan_n_out(const smilehip_plan *p) { return plan_n_static(p) * (1 + p->cfg.n_delta); }
static int plan_row_extra(const smilehip_plan *p) { return p->cfg.chain_kind == SMILEHIP_CHAIN_IS09 ? p->cfg.sma_win / 2 : 0; }

extern "C" int smilehip_plan_geometry(const smilehip_plan *p, smilehip_geometry *g) {
  if (!p || !g) return fail(SMILEHIP_ERR_INVALID, "smilehip_plan_geometry: null argument");
  g->frame_size = p->geo.N;
  g->frame_step = p->geo.H;
  g->fft_size = p->geo.Nfft;
  g->n_bins = p->geo.K;
  g->n_static = plan_n_static(p);
  g->n_out = plan_n_out(p);
  g->frame_period = p->geo.frame_period;
  g->fft_frame_size_sec = p->geo.fft_frame_size_sec;
  return SMILEHIP_OK;
}

extern "C" int64_t smilehip_num_frames(const smilehip_plan *p, int64_t n_samples) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_num_frames: null plan");
  if (n_samples < p->geo.N) return 0;
  return (n_samples - p->geo.N) / p->geo.H + 1;
}

extern "C" double smilehip_frame_time(const smilehip_plan *p, int64_t t) {
  // vIdx * level period; the framer's level period is frameStep
  // (winToVecProcessor.cpp:562-568), time = vIdx*T via squashTimeMeta
  return p ? double(t) * p->geo.frame_period : 0.0;
}

extern "C" double smilehip_row_time(const smilehip_plan *plan, int64_t n_frames, int64_t row) {
  if (!plan || row < 0) return 0.0;
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || plan->cfg.chain_kind == SMILEHIP_CHAIN_PLP || n_frames <= 1)
    return smilehip_frame_time(plan, row);
  return smilehip_frame_time(plan, row < n_frames - 1 ? row : n_frames - 1);
}

template <typename T>
static int64_t copy_out(const std::vector<T> &v, T *out, int64_t cap) {
  if (out) {
    if (cap < (int64_t)v.size()) return fail(SMILEHIP_ERR_INVALID, "output buffer too small");
    std::memcpy(out, v.data(), v.size() * sizeof(T));
  }
  return (int64_t)v.size();
}
extern "C" int64_t smilehip_plan_get_window(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->h_window, o, cap); }
extern "C" int64_t smilehip_plan_get_mel_weights(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->mel.coef, o, cap); }
extern "C" int64_t smilehip_plan_get_mel_chanmap(const smilehip_plan *p, int32_t *o, int64_t cap) { return copy_out(p->mel.chan, o, cap); }
extern "C" int64_t smilehip_plan_get_dct(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->dct.cos_rows, o, cap); }
extern "C" int64_t smilehip_plan_get_lifter(const smilehip_plan *p, float *o, int64_t cap) { return copy_out(p->dct.lifter, o, cap); }

// ------------------------------------------------------------------ batch
extern "C" int smilehip_batch_create(smilehip_plan *plan, const int64_t *h_off, int32_t n_utt, smilehip_batch **out) {
  if (!plan || !out || n_utt < 0 || (n_utt > 0 && !h_off)) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_create: bad argument");
  *out = nullptr;
  if (!plan->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached (tables only)");
  if (plan->stage_mask != SMILEHIP_STAGE_ALL) return fail(SMILEHIP_ERR_INVALID, "single-component plan cannot run the fused chain");
  HIP_TRY(hipSetDevice(plan->ctx->device));
  auto *b = new (std::nothrow) smilehip_batch();
  if (!b) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  b->plan = plan;
  b->n_utt = n_utt;
  b->h_samp_off.assign(h_off, h_off + (n_utt ? n_utt + 1 : 0));
  if (n_utt == 0) b->h_samp_off.assign(1, 0);
  b->h_frame_off.assign(size_t(n_utt) + 1, 0);
  b->h_row_off.assign(size_t(n_utt) + 1, 0);
  const int short_T = chain_short_max();
  const int row_extra = plan_row_extra(plan);
  std::vector<int32_t> tile_utt, tile_t0, dtile_utt, dtile_t0, run_utt, run_t0;
  std::vector<TileRec> tile_rec;
  const int64_t dtile = chain_tile_rows();
  const int64_t tile_frames = plan->use_fast ? fast512_tile_frames() : (int64_t(1) << 40);
  for (int32_t u = 0; u < n_utt; ++u) {
    const int64_t len = h_off[u + 1] - h_off[u];
    if (len < 0) {
      delete b;
      return fail(SMILEHIP_ERR_INVALID, "sample offsets must be non-decreasing (utterance %d)", u);
    }
    const int64_t T = smilehip_num_frames(plan, len);
    int64_t rows = T > 0 ? T + row_extra : 0;
    if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB) {
      // rows = T60 + 1 where T60 = frames of the 60 ms framer ([is13_frame60]); none if T60 < 4
      const int64_t N60 = std::lround(0.060 / plan->geo.period);
      const int64_t T60 = (len >= N60) ? (len - N60) / plan->geo.H + 1 : 0;
      rows = (T60 >= 4) ? T60 + 1 : 0;
      for (int64_t t0 = 0; t0 < T; t0 += compare_run_frames()) {
        run_utt.push_back(u);
        run_t0.push_back((int32_t)t0);
      }
    }
    b->h_frame_off[u + 1] = b->h_frame_off[u] + T;
    b->h_row_off[u + 1] = b->h_row_off[u] + rows;
    if (T > 0 && T <= short_T) b->h_short.push_back(u);
    if (T > 0 && (h_off[u] & 1)) b->all_even = false;
    for (int64_t t0 = 0; t0 < T; t0 += tile_frames) {
      tile_utt.push_back(u);
      tile_t0.push_back((int32_t)t0);
      TileRec r;
      r.samp0 = h_off[u] + t0 * plan->geo.H;
      r.row0 = b->h_frame_off[u] + t0;
      r.n_frames = (int32_t)std::min<int64_t>(tile_frames, T - t0);
      r.pad = 0;
      tile_rec.push_back(r);
    }
    for (int64_t t0 = 0; t0 < rows; t0 += dtile) {
      dtile_utt.push_back(u);
      dtile_t0.push_back((int32_t)t0);
    }
  }
  b->total_frames = b->h_frame_off[n_utt];
  b->total_rows = b->h_row_off[n_utt];
  b->n_tiles = (int32_t)tile_utt.size();
  b->n_dtiles = (int32_t)dtile_utt.size();
  int rc;
  if ((rc = b->d_samp_off.upload(b->h_samp_off)) || (rc = b->d_frame_off.upload(b->h_frame_off)) ||
      (rc = b->d_row_off.upload(b->h_row_off)) ||
      (rc = b->d_tile_utt.upload(tile_utt)) || (rc = b->d_tile_t0.upload(tile_t0)) || (rc = b->d_tile_rec.upload(tile_rec)) ||
      (rc = b->d_dtile_utt.upload(dtile_utt)) || (rc = b->d_dtile_t0.upload(dtile_t0)) ||
      (rc = b->d_short.upload(b->h_short))) {
    delete b;
    return rc;
  }
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB) {
    b->n_runs = (int32_t)run_utt.size();
    if ((rc = b->d_run_utt.upload(run_utt)) || (rc = b->d_run_t0.upload(run_t0))) {
      delete b;
      return rc;
    }
    const size_t nf = size_t(b->total_frames ? b->total_frames : 1);
    if (hipMalloc(reinterpret_cast<void **>(&b->d_rawA.p), nf * 4 * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&b->d_rawB.p), nf * 55 * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&b->d_mel1.p), nf * 26 * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the ComParE scratch matrices failed");
    }
    b->d_rawA.n = nf * 4; b->d_rawB.n = nf * 55; b->d_mel1.n = nf * 26;
    (void)hipMemset(b->d_rawA.p, 0, nf * 4 * sizeof(float));
  }
  if ((plan->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || plan->cfg.chain_kind == SMILEHIP_CHAIN_PLP) && plan->cfg.n_delta > 0 &&
      plan->ctx && b->total_frames > 0) {
    const size_t n = size_t(b->total_frames) * size_t(plan->dct.n_mfcc);
    if (hipMalloc(reinterpret_cast<void **>(&b->d_static.p), n * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the static-block scratch failed");
    }
    b->d_static.n = n;
  }
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_IS09) {
    std::vector<float> zero;   // allocate only
    b->d_raw16.release();
    b->d_raw16.n = size_t(b->total_frames) * 16;
    if (hipMalloc(reinterpret_cast<void **>(&b->d_raw16.p), (b->d_raw16.n ? b->d_raw16.n : 1) * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the IS09 scratch matrix failed");
    }
  }
  *out = b;
  return SMILEHIP_OK;
}

extern "C" void smilehip_batch_destroy(smilehip_batch *b) { delete b; }
extern "C" int64_t smilehip_batch_total_frames(const smilehip_batch *b) { return b ? b->total_frames : 0; }
extern "C" int64_t smilehip_batch_total_rows(const smilehip_batch *b) { return b ? b->total_rows : 0; }
extern "C" int smilehip_batch_frame_offsets(const smilehip_batch *b, int64_t *o) {
  if (!b || !o) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_frame_offsets: null argument");
  std::memcpy(o, b->h_row_off.data(), b->h_row_off.size() * sizeof(int64_t));
  return SMILEHIP_OK;
}

// -------------------------------------------------------------------- run
static void fill_params(const smilehip_plan *p, const smilehip_batch *b, const int16_t *d_pcm, float *d_out,
                        int64_t ld, LldParams &P) {
  std::memset(&P, 0, sizeof(P));
  P.pcm = d_pcm;
  P.pcm_total = b->h_samp_off.back();
  P.samp_off = b->d_samp_off.p;
  P.frame_off = b->d_frame_off.p;
  P.tile_utt = b->d_tile_utt.p;
  P.tile_t0 = b->d_tile_t0.p;
  P.tile_rec = b->d_tile_rec.p;
  P.n_utt = b->n_utt;
  P.n_tiles = b->n_tiles;
  P.total_frames = b->total_frames;
  P.out = d_out;
  P.ld_out = ld;
  P.N = (int32_t)p->geo.N;
  P.H = (int32_t)p->geo.H;
  P.Nfft = (int32_t)p->geo.Nfft;
  P.K = (int32_t)p->geo.K;
  P.pad_left = p->cfg.zero_pad_symmetric ? (int32_t)((p->geo.Nfft - p->geo.N) / 2) : 0;
  P.preemph = p->cfg.preemph;
  P.de = p->cfg.preemph_de;
  P.k = p->cfg.preemph_k;
  P.one_minus_k = 1 - p->cfg.preemph_k;     // (1-k) in float, vectorPreemphasis.cpp:94
  P.win_offset = (float)p->cfg.win_offset;
  P.window = p->d_window.p;
  P.tw_half = p->d_tw_half.p;
  P.tw_full = p->d_tw_full.p;
  P.mel_coef = p->d_mel_coef.p;
  P.mel_rng = p->d_mel_rng.p;
  P.mel_scale = p->mel.scale;
  P.use_power = p->cfg.use_power;
  P.n_bands = p->mel.n_bands;
  P.dct_rows = p->d_dct_rows.p;
  P.dct_gain = p->d_dct_gain.p;
  P.n_mfcc = p->dct.n_mfcc;
  P.melfloor = p->dct.melfloor;
  P.log_floor = p->dct.log_floor;
  P.plp = p->cfg.chain_kind == SMILEHIP_CHAIN_PLP;
  P.plp_order = p->cfg.plp_lp_order;
  P.plp_compression = p->cfg.plp_compression;
  P.plp_eql = p->d_plp_eql.p;
  P.plp_cos = p->d_plp_cos.p;
  P.plp_sin = p->d_plp_sin.p;
}

// R13 for a batch whose rows == frames: level 0 = x (leading dimension ld_x); writes [copy of x at copy_col (if >= 0) |
// order 1 at D | order 2 at 2D] into out
static int delta_chain_from(smilehip_plan *plan, smilehip_batch *b, const float *d_x, int64_t ld_x, int copy_col, float *d_io,
                            int64_t ld, int32_t D, int32_t W, int32_t n_orders, void *stream) {
  if (!plan || !b || !d_io) return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_chain: null argument");
  if (n_orders < 1 || n_orders > 2 || W < 1 || W > 4 || D < 1 || D > 16 || ld < (int64_t)D * (1 + n_orders))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_chain: unsupported D=%d W=%d orders=%d ld=%lld", D, W, n_orders, (long long)ld);
  if (b->total_frames == 0) return SMILEHIP_OK;
  // rows == frames is what this entry point assumes (d_io holds the static block)
  if (b->total_rows != b->total_frames) return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_chain: batch belongs to a chain with extra rows");
  ChainParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.frame_off = b->d_frame_off.p;
  Q.row_off = b->d_row_off.p;
  Q.tile_utt = b->d_dtile_utt.p;
  Q.tile_t0 = b->d_dtile_t0.p;
  Q.n_tiles = b->n_dtiles;
  Q.n_utt = b->n_utt;
  Q.x = d_x;
  Q.ld_x = ld_x;
  Q.copy_col = copy_col;
  Q.out = d_io;
  Q.ld_out = ld;
  Q.D = D;
  Q.n_stages = n_orders;
  Q.kind[0] = Q.kind[1] = 0;
  Q.W[0] = Q.W[1] = W;
  Q.out_col[0] = D;
  Q.out_col[1] = 2 * D;
  Q.short_T = chain_short_max();
  Q.short_utts = b->d_short.p;
  Q.n_short = (int32_t)b->h_short.size();
  hipError_t e = launch_chain(Q, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "window-chain kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_delta_chain(smilehip_plan *plan, smilehip_batch *b, float *d_io, int64_t ld, int32_t D,
                                    int32_t W, int32_t n_orders, void *stream) {
  return delta_chain_from(plan, b, d_io, ld, -1, d_io, ld, D, W, n_orders, stream);
}

extern "C" int smilehip_mfcc_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out,
                                 int64_t ld_out, void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run: plan/batch mismatch");
  if (plan->cfg.chain_kind != SMILEHIP_CHAIN_MFCC && plan->cfg.chain_kind != SMILEHIP_CHAIN_PLP)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run: plan is not an MFCC / PLP chain (use smilehip_lld_run)");
  const int n_out = plan->dct.n_mfcc * (1 + plan->cfg.n_delta);
  if (ld_out < n_out) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < n_out %d", (long long)ld_out, n_out);
  if (b->total_frames == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run: null device pointer");
  hipStream_t s = (hipStream_t)stream;
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  // With deltas to follow, the static block goes to a compact [frames x n_mfcc] scratch: the frame kernel
  // then writes whole lines, the window-chain kernel reads 1/3 of what it would read from 39-float rows,
  // and writes every output row in one piece (static | delta | accel).
  const bool compact = plan->cfg.n_delta > 0 && b->d_static.p != nullptr;
  if (compact) {
    P.out = b->d_static.p;
    P.ld_out = plan->dct.n_mfcc;
  }
  hipEvent_t *ev = plan->ev[plan->n_timed % smilehip_plan::kRing];
  if (plan->timing) {
    for (int i = 0; i < 3; ++i)
      if (!ev[i]) HIP_TRY(hipEventCreate(&ev[i]));
    HIP_TRY(hipEventRecord(ev[0], s));
  }
  hipError_t e;
  if (plan->use_fast) {
    Fast512Tables F;
    F.tw256 = plan->d_tw256.p;
    F.tw512 = plan->d_tw512.p;
    F.win = plan->d_fwin.p;
    F.melw = plan->d_melw.p;
    F.melo = plan->d_melo.p;
    F.dct28 = plan->d_dct28.p;
    F.band_slots = plan->d_band_slots.p;
    F.mel_units = plan->fast.mel_units;
    F.n_slots = plan->fast.n_slots;
    F.stage_floats = plan->fast.stage_floats;
    F.stage_alloc = plan->fast.stage_alloc;
    F.mel_scale = plan->fast.mel_scale;
    F.plp_eql = plan->d_plp_eql.p;
    F.plp_sin = plan->d_plp_sin.p;
    const bool aligned = b->all_even && ((reinterpret_cast<uintptr_t>(d_pcm) & 3) == 0);
    e = launch_mfcc512(P, F, plan->fast, aligned, s);
  } else {
    e = launch_mfcc_generic(P, s);
  }
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "mfcc kernel launch failed: %s", hipGetErrorString(e));
  if (plan->timing) HIP_TRY(hipEventRecord(ev[1], s));
  if (plan->cfg.n_delta > 0) {
    int rc = compact ? delta_chain_from(plan, b, b->d_static.p, plan->dct.n_mfcc, 0, d_out, ld_out, plan->dct.n_mfcc,
                                        plan->cfg.delta_win, plan->cfg.n_delta, stream)
                     : smilehip_delta_chain(plan, b, d_out, ld_out, plan->dct.n_mfcc, plan->cfg.delta_win, plan->cfg.n_delta, stream);
    if (rc) return rc;
  }
  if (plan->timing) {
    HIP_TRY(hipEventRecord(ev[2], s));
    plan->n_timed++;
  }
  return SMILEHIP_OK;
}

// IS09 LLD set: frame kernel -> pitch smoother -> SMA + delta chain
static int is09_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out, void *stream) {
  const int n_out = plan_n_out(plan);
  if (ld_out < n_out) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < n_out %d", (long long)ld_out, n_out);
  if (b->total_frames == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: null device pointer");
  hipStream_t s = (hipStream_t)stream;
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  Is09Params I;
  I.raw16 = b->d_raw16.p;
  I.fsSec = (float)plan->geo.fft_frame_size_sec;
  I.maxPitch = plan->cfg.pitch_max;
  I.voicingCutoff = plan->cfg.voicing_cutoff;
  if (I.voicingCutoff > 1.0) I.voicingCutoff = 1.0;       // pitchACF.cpp:96-98
  if (I.voicingCutoff < 0.0) I.voicingCutoff = 0.0;
  if (I.maxPitch < 0.0) I.maxPitch = 0.0;
  hipError_t e = launch_is09(P, I, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "IS09 kernel launch failed: %s", hipGetErrorString(e));
  ChainParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.frame_off = b->d_frame_off.p;
  Q.row_off = b->d_row_off.p;
  Q.tile_utt = b->d_dtile_utt.p;
  Q.tile_t0 = b->d_dtile_t0.p;
  Q.n_tiles = b->n_dtiles;
  Q.n_utt = b->n_utt;
  Q.x = b->d_raw16.p;
  Q.ld_x = 16;
  Q.copy_col = -1;
  Q.out = d_out;
  Q.ld_out = ld_out;
  Q.D = 16;
  Q.n_stages = 2;
  Q.kind[0] = 1; Q.W[0] = plan->cfg.sma_win / 2;          // cContourSmoother
  Q.kind[1] = 0; Q.W[1] = plan->cfg.delta_win;            // cDeltaRegression
  Q.out_col[0] = 0;
  Q.out_col[1] = 16;
  Q.short_T = chain_short_max();
  Q.short_utts = b->d_short.p;
  Q.n_short = (int32_t)b->h_short.size();
  e = launch_chain(Q, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "window-chain kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

// ComParE groups A+B: frame kernel -> RASTA scan -> group A (multi-length SMA+delta) + group B chain
static int compare_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out, void *stream) {
  const int n_out = plan_n_out(plan);
  if (ld_out < n_out) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < n_out %d", (long long)ld_out, n_out);
  if (b->total_frames == 0 || b->total_rows == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: null device pointer");
  hipStream_t s = (hipStream_t)stream;
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  CompareParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.run_utt = b->d_run_utt.p;
  Q.run_t0 = b->d_run_t0.p;
  Q.rawA = b->d_rawA.p;
  Q.rawB = b->d_rawB.p;
  Q.mel1 = b->d_mel1.p;
  Q.eql = plan->d_eql.p;
  Q.eql_log = plan->d_eql_log.p;
  Q.sharp_w = plan->d_sharp.p;
  Q.plp_melfloor = 0.00000000093f;     // cPlp melfloor default (plp.cpp:66), htkcompatible = 0
  Q.compression = 0.33f;
  Q.rasta_iir = plan->rasta_iir;
  for (int i = 0; i < 5; ++i) Q.rasta_fir[i] = plan->rasta_fir[i];
  Q.fsSec = plan->geo.fft_frame_size_sec;
  Q.N60 = (int32_t)std::lround(0.060 / plan->geo.period);
  for (int i = 0; i < 2; ++i) {
    Q.band_iL[i] = plan->band_iL[i]; Q.band_iR[i] = plan->band_iR[i];
    Q.band_wL[i] = plan->band_wL[i]; Q.band_wR[i] = plan->band_wR[i];
  }
  Q.slope_Sf = plan->slope_Sf;
  Q.slope_S2f = plan->slope_S2f;
  hipError_t e = launch_compare(P, Q, b->n_runs, b->d_row_off.p, b->total_rows, d_out, ld_out, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "ComParE kernel launch failed: %s", hipGetErrorString(e));
  ChainParams C;
  std::memset(&C, 0, sizeof(C));
  C.frame_off = b->d_frame_off.p;
  C.row_off = b->d_row_off.p;
  C.tile_utt = b->d_dtile_utt.p;
  C.tile_t0 = b->d_dtile_t0.p;
  C.n_tiles = b->n_dtiles;
  C.n_utt = b->n_utt;
  C.x = b->d_rawB.p;
  C.ld_x = 55;
  C.copy_col = -1;
  C.out = d_out;
  C.ld_out = ld_out;
  C.D = 55;
  C.n_stages = 2;
  C.kind[0] = 1; C.W[0] = 1;
  C.kind[1] = 0; C.W[1] = 2;
  C.out_col[0] = 4;
  C.out_col[1] = 59 + 4;
  C.short_T = chain_short_max();
  C.short_utts = b->d_short.p;
  C.n_short = (int32_t)b->h_short.size();
  e = launch_chain(C, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "window-chain kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

// ------------------------------------------------------------- functionals
extern "C" uint32_t smilehip_functionals_is09_mask(void) {
  return SMILEHIP_FUNC_MAX | SMILEHIP_FUNC_MIN | SMILEHIP_FUNC_RANGE | SMILEHIP_FUNC_MAXPOS | SMILEHIP_FUNC_MINPOS |
         SMILEHIP_FUNC_AMEAN | SMILEHIP_FUNC_LINREGC1 | SMILEHIP_FUNC_LINREGC2 | SMILEHIP_FUNC_LINREGERRQ |
         SMILEHIP_FUNC_STDDEV | SMILEHIP_FUNC_SKEWNESS | SMILEHIP_FUNC_KURTOSIS;
}

extern "C" int smilehip_functionals_count(uint32_t mask) {
  if (mask & ~SMILEHIP_FUNC_ALL) return -1;
  return __builtin_popcount(mask);
}

static const int kIs09FuncRowsCut = 3;      // rows = T+1; functionals see max(1, T-2)

extern "C" int smilehip_batch_func_rows(const smilehip_batch *b, int64_t *rows) {
  if (!b || !rows) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_func_rows: null argument");
  if (b->plan->cfg.chain_kind != SMILEHIP_CHAIN_IS09)
    return fail(SMILEHIP_ERR_INVALID, "functionals are defined for IS09 chain plans only");
  for (int32_t u = 0; u < b->n_utt; ++u) {
    const int64_t r = b->h_row_off[u + 1] - b->h_row_off[u];
    rows[u] = r > 0 ? std::max<int64_t>(1, r - kIs09FuncRowsCut) : 0;
  }
  return SMILEHIP_OK;
}

extern "C" int smilehip_batch_functionals(smilehip_plan *plan, smilehip_batch *b, const float *d_lld, int64_t ld_lld,
                                          uint32_t mask, float *d_func, int64_t ld_func, void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals: plan/batch mismatch");
  if (plan->cfg.chain_kind != SMILEHIP_CHAIN_IS09)
    return fail(SMILEHIP_ERR_INVALID, "functionals are defined for IS09 chain plans only");
  const int per = smilehip_functionals_count(mask);
  if (per <= 0) return fail(SMILEHIP_ERR_INVALID, "invalid functionals mask 0x%x", mask);
  const int n_cols = plan_n_out(plan);
  if (ld_lld < n_cols || ld_func < (int64_t)n_cols * per)
    return fail(SMILEHIP_ERR_INVALID, "leading dimensions too small (ld_lld %lld, ld_func %lld)", (long long)ld_lld,
                (long long)ld_func);
  if (b->n_utt == 0) return SMILEHIP_OK;
  if (!d_func || (!d_lld && b->total_rows > 0)) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals: null device pointer");
  FuncParams P;
  std::memset(&P, 0, sizeof(P));
  P.row_off = b->d_row_off.p;
  P.x = d_lld;
  P.ld_x = ld_lld;
  P.n_cols = n_cols;
  P.rows_cut = kIs09FuncRowsCut;
  P.single_rows = -1;
  P.mask = mask;
  P.out = d_func;
  P.ld_out = ld_func;
  hipError_t e = launch_functionals(P, b->n_utt, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "functionals kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_functionals_matrix(smilehip_context *ctx, const float *d_x, int64_t ld_x, int64_t rows, int32_t cols,
                                           uint32_t mask, float *d_out, void *stream) {
  const int per = smilehip_functionals_count(mask);
  if (!ctx || per <= 0 || rows < 1 || cols < 1 || ld_x < cols || !d_x || !d_out)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_functionals_matrix: bad argument");
  FuncParams P;
  std::memset(&P, 0, sizeof(P));
  P.x = d_x;
  P.ld_x = ld_x;
  P.n_cols = cols;
  P.mask = mask;
  P.single_rows = rows;
  P.out = d_out;
  P.ld_out = (int64_t)cols * per;
  hipError_t e = launch_functionals(P, 1, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "functionals kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_lld_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out,
                                void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: plan/batch mismatch");
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || plan->cfg.chain_kind == SMILEHIP_CHAIN_PLP)
    return smilehip_mfcc_run(plan, b, d_pcm, d_out, ld_out, stream);
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB) return compare_run(plan, b, d_pcm, d_out, ld_out, stream);
  return is09_run(plan, b, d_pcm, d_out, ld_out, stream);
}

extern "C" int smilehip_lld_run_host(smilehip_plan *plan, smilehip_batch *b, const int16_t *h_pcm, int64_t n_samples,
                                     float *h_out) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run_host: plan/batch mismatch");
  if (n_samples < b->h_samp_off.back()) return fail(SMILEHIP_ERR_INVALID, "PCM buffer shorter than the batch layout");
  if (b->total_rows == 0) return SMILEHIP_OK;
  if (!h_pcm || !h_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run_host: null pointer");
  HIP_TRY(hipSetDevice(plan->ctx->device));
  const int n_out = plan_n_out(plan);
  int16_t *d_pcm = nullptr;
  float *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_pcm, size_t(n_samples) * sizeof(int16_t)));
  hipError_t e = hipMalloc((void **)&d_out, size_t(b->total_rows) * n_out * sizeof(float));
  if (e != hipSuccess) {
    (void)hipFree(d_pcm);
    return fail(SMILEHIP_ERR_HIP, "hipMalloc(out) failed: %s", hipGetErrorString(e));
  }
  int rc = SMILEHIP_OK;
  e = hipMemcpy(d_pcm, h_pcm, size_t(n_samples) * sizeof(int16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = smilehip_lld_run(plan, b, d_pcm, d_out, n_out, nullptr);
    if (rc == SMILEHIP_OK) e = hipMemcpy(h_out, d_out, size_t(b->total_rows) * n_out * sizeof(float), hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_pcm);
  (void)hipFree(d_out);
  if (rc) return rc;
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "HIP copy failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_mfcc_run_host(smilehip_plan *plan, smilehip_batch *b, const int16_t *h_pcm, int64_t n_samples,
                                      float *h_out) {
  if (!plan || plan->cfg.chain_kind != SMILEHIP_CHAIN_MFCC) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run_host: plan is not an MFCC chain");
  return smilehip_lld_run_host(plan, b, h_pcm, n_samples, h_out);
}

extern "C" int smilehip_plan_set_timing(smilehip_plan *plan, int enable) {
  if (!plan) return fail(SMILEHIP_ERR_INVALID, "null plan");
  plan->timing = enable != 0;
  plan->n_timed = 0;
  return SMILEHIP_OK;
}

// Average over the runs recorded since set_timing (at most the last kRing).
// The caller must have synchronised the stream.
extern "C" int smilehip_plan_last_timing(smilehip_plan *plan, float *ms_main, float *ms_delta) {
  if (!plan || plan->n_timed <= 0) return fail(SMILEHIP_ERR_INVALID, "no timing recorded");
  const int64_t n = plan->n_timed < smilehip_plan::kRing ? plan->n_timed : smilehip_plan::kRing;
  double sa = 0.0, sd = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    float a = 0.f, d = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, plan->ev[i][0], plan->ev[i][1]));
    HIP_TRY(hipEventElapsedTime(&d, plan->ev[i][1], plan->ev[i][2]));
    sa += a;
    sd += d;
  }
  if (ms_main) *ms_main = float(sa / double(n));
  if (ms_delta) *ms_delta = float(sd / double(n));
  return SMILEHIP_OK;
}

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
                      abs_cepstrum, p->d_tw_half.p, p->d_tw_full.p, (hipStream_t)stream), "acf");
}

extern "C" int smilehip_pitchacf_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t n, int64_t n_frames,
                                        double fs_sec, double max_pitch, double *d_voicing, int32_t *d_max_idx, void *stream) {
  if (!ctx || n < 4 || n_frames < 0 || ld_src < 2 * n || (n_frames > 0 && (!d_src || !d_voicing || !d_max_idx)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_pitchacf_frames: bad argument");
  STAGE_RET(stage_pitchacf(d_src, ld_src, n_frames, (int)n, fs_sec, max_pitch, d_voicing, d_max_idx, (hipStream_t)stream), "pitchacf");
}

extern "C" int smilehip_spectral_frames(smilehip_plan *p, const float *d_mag, int64_t ld_src, float *d_state, int first,
                                        float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  if (!p->d_sharp.p) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: plan was built without SMILEHIP_STAGE_SPECTRAL");
  if (p->geo.K != 257) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: K = %lld, the kernel covers K = 257", (long long)p->geo.K);
  if (!d_state && n_frames > 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_spectral_frames: null state buffer");
  int rc = check_frames(d_mag, d_dst, ld_src, ld_dst, n_frames, p->geo.K, 15, "smilehip_spectral_frames");
  if (rc) return rc;
  SpectralConsts C;
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
  STAGE_RET(stage_plp(d_mel, ld_src, n_bands, d_eql, Q, new_rasta != 0, d_state, d_dst, ld_dst, n_frames, (hipStream_t)stream), "plp");
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

extern "C" int smilehip_window_op_row(smilehip_context *ctx, const float *d_x, float *d_y, int64_t n_t, int kind, int W,
                                      void *stream) {
  if (!ctx || n_t < 0 || W < 1 || (kind != 0 && kind != 1) || (n_t > 0 && (!d_x || !d_y)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_window_op_row: bad argument");
  STAGE_RET(stage_window_op(d_x, d_y, n_t, kind, W, delta_norm(W), (hipStream_t)stream), "window_op");
}

extern "C" int smilehip_fftmag_frames(smilehip_plan *p, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                      int64_t n_frames, void *stream) {
  if (!p) return fail(SMILEHIP_ERR_INVALID, "smilehip_fftmag_frames: null plan");
  if (!p->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached");
  int rc = check_frames(d_src, d_dst, ld_src, ld_dst, n_frames, p->geo.Nfft, p->geo.K, "smilehip_fftmag_frames");
  if (rc) return rc;
  STAGE_RET(stage_fftmag(d_src, ld_src, d_dst, ld_dst, n_frames, (int)p->geo.Nfft, (hipStream_t)stream), "fftmag");
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
